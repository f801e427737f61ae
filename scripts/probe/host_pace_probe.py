#!/usr/bin/env python3
"""Where does the host wait inside validate()?  Times JitterStream.slice and each C-ABI call on the host (no device sync added)."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402
from gens_amd.config import gens_model_conf  # noqa: E402
import gens_amd.models.modules.implicit_surface as M  # noqa: E402

dev = torch.device("cuda:0")
dims = [256, 128, 64]
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
imgs, intrs, c2ws = sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev)
feats = [f.to(dev) for f in sc["features"]]
vols = [v.to(dev) for v in synthetic.make_volumes(dims, seed=100)]
ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640)
ro, rd = ro.to(dev), rd.to(dev)
near, far = sc["near"].to(dev), sc["far"].to(dev)
torch.manual_seed(0)
surf = M.ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"]).to(dev).eval()
surf.val_chunk = 32768
T = time.perf_counter
acc = collections.defaultdict(float)
orig_slice = M.JitterStream.slice


def timed_slice(self, s, e):
    t = T()
    r = orig_slice(self, s, e)
    acc["jitter.slice"] += T() - t
    return r


M.JitterStream.slice = timed_slice
orig_call = L.call


def timed_call(name, *a, **k):
    t = T()
    r = orig_call(name, *a, **k)
    acc["abi:" + name] += T() - t
    return r


L.call = timed_call
ops.L.call = timed_call
for it in range(3):
    acc.clear()
    with torch.no_grad():
        _, masks = ops.volume_build(feats[:3], intrs, c2ws, dims)
        scene = M.Scene(vols, masks, imgs, feats, feats, intrs, c2ws)
        torch.cuda.synchronize()
        t0 = T()
        surf.validate(ro, rd, near, far, vols, masks, imgs, feats, feats, intrs, c2ws, None, None, (480, 640), extract_geometry=False, scene=scene)
        t1 = T()
    print(f"validate {1e3 * (t1 - t0):.1f} ms; host time inside: " + ", ".join(f"{k} {1e3 * v:.1f}" for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:8]))

import cProfile
import pstats
with torch.no_grad():
    _, masks = ops.volume_build(feats[:3], intrs, c2ws, dims)
    scene = M.Scene(vols, masks, imgs, feats, feats, intrs, c2ws)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    surf.validate(ro, rd, near, far, vols, masks, imgs, feats, feats, intrs, c2ws, None, None, (480, 640), extract_geometry=False, scene=scene)
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
