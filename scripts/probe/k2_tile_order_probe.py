"""Does K2's forward gain from a ray-tile point order (neighbouring rays at the same depth share texel lines in L1)?  The same 4.19 M points of
32 768 rays x 128 samples in ray-major order (what the callers pass) and permuted into tiles of R rays x T samples."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
L.load()
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
feats = [f.to(dev) for f in sc["features"]]
intrs, c2ws = sc["intrs"].to(dev), sc["c2ws"].to(dev)
dims = [256, 128, 64]
vols = [v.to(dev) for v in synthetic.make_volumes(dims, seed=1)]
b, n = 32768, 128
ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640)
ro, rd = ro[:b].to(dev).contiguous(), rd[:b].to(dev).contiguous()
near, far = sc["near"].to(dev), sc["far"].to(dev)


def timed(fn, reps=40, rounds=5):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / reps * 1e3)
    return statistics.median(out)


with torch.no_grad():
    _, masks = ops.volume_build(feats[:3], intrs, c2ws, dims)
    mset = ops.VolumeSet.masks(masks)
    vpack = ops.VolumeSet.packed(vols)
    z = (near + (far - near) * torch.linspace(0, 1, n, device=dev)[None]).expand(b, n).contiguous()
    pts, _ = ops.ray_points(ro, rd, z, mset, mid=True, sample_dist=1 / 32)
    print(f"ray-major: {timed(lambda: ops.lookup_volume(pts, vpack)):8.1f} us")
    p4 = pts.view(b, n, 3)
    for r, t in ((16, 16), (8, 32), (4, 64), (32, 8), (64, 4), (8, 8), (16, 4), (4, 16)):
        q = p4.view(b // r, r, n // t, t, 3).permute(0, 2, 1, 3, 4).contiguous().view(-1, 3)
        print(f"tiles of {r:3d} rays x {t:3d} samples: {timed(lambda: ops.lookup_volume(q, vpack)):8.1f} us")
        q = p4.view(b // r, r, n // t, t, 3).permute(0, 2, 3, 1, 4).contiguous().view(-1, 3)
        print(f"tiles of {t:3d} samples x {r:3d} rays (ray fastest): {timed(lambda: ops.lookup_volume(q, vpack)):8.1f} us")
