"""Workload for the PMC passes of scripts/probe/gather_traffic.sh: K2 forward (packed, 4.19 M ray samples, three levels) and K4 forward (3.8 M points x 4
source views, five levels), a few launches each, same inputs as scripts/kernel_bench.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
L.load()
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
imgs, intrs, c2ws = sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev)
feats = [f.to(dev) for f in sc["features"]]
dims = [256, 128, 64]
vols = [v.to(dev) for v in synthetic.make_volumes(dims, seed=1)]
b = 32768
ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640)
ro, rd = ro[:b].to(dev).contiguous(), rd[:b].to(dev).contiguous()
near, far = sc["near"].to(dev), sc["far"].to(dev)
with torch.no_grad():
    _, masks = ops.volume_build(feats[:3], intrs, c2ws, dims)
    mset = ops.VolumeSet.masks(masks)
    vpack = ops.VolumeSet.packed(vols)
    z128 = (near + (far - near) * torch.linspace(0, 1, 128, device=dev)[None]).expand(b, 128).contiguous()
    pts, valid = ops.ray_points(ro, rd, z128, mset, mid=True, sample_dist=1 / 32)
    views = ops.SceneViews(imgs, intrs, c2ws, feats)
    pv = pts[valid.bool()].contiguous()
    for _ in range(5):
        ops.lookup_volume(pts, vpack)
        ops.lookup_feature(pv, views)
    torch.cuda.synchronize()
print("points", pts.shape[0], "valid", pv.shape[0])
