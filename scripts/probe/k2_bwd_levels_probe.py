#!/usr/bin/env python3
"""K2 backward (gens_lookup_volume_bwd, one lane per point): where does the time of the volume scatter go, level by level?  1 M points
(ray samples of a chunk, and uniformly random ones), volumes 256^3 / 128^3 / 64^3, the gradient buffers of a subset of the levels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
dims = [256, 128, 64]
vols = [v.to(dev) for v in synthetic.make_volumes(dims, seed=1)]
n = 1 << 20
g = torch.Generator().manual_seed(0)
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640)
ro, rd = ro[:8192].to(dev), rd[:8192].to(dev)
z = torch.linspace(float(sc["near"].min()), float(sc["far"].max()), 128, device=dev)
ray_pts = (ro[:, None] + rd[:, None] * z[None, :, None]).reshape(-1, 3)[:n].contiguous()
cases = {"ray samples": ray_pts, "uniform": (torch.rand(n, 3, generator=g) * 2 - 1).to(dev)}
for layout_name, layout in (("planar", L.LAYOUT_PLANAR), ("packed", L.LAYOUT_PACKED)):
    vs = ops.VolumeSet.packed(vols) if layout == L.LAYOUT_PACKED else ops._vset(layout, vols)
    for name, pts in cases.items():
        g_out = torch.randn(n, 4 * len(dims), device=dev)
        g_pts = torch.empty(n, 3, device=dev)
        for subset, per_point in (([], False), ([0], False), ([1], False), ([2], False), ([0, 1, 2], False), ([0, 1, 2], True)):
            os.environ.pop("GENS_K2_SCATTER_PER_POINT", None)
            if per_point:
                os.environ["GENS_K2_SCATTER_PER_POINT"] = "1"              # the round-1 form: one lane per point, the scatter inside the per-point kernel
            grads = [torch.zeros_like(t) if l in subset else None for l, t in enumerate(vs.tensors)]
            fn = lambda: L.call("gens_lookup_volume_bwd", vs.table, vs.dim_table, vs.n, layout, L.ptr(pts), L.ptr(g_out), n, L.ptr_table(grads),  # noqa: E731
                                L.ptr(g_pts), L.stream())
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record()
            torch.cuda.synchronize()
            print(f"{layout_name:7s} {name:12s} gradient levels {str(subset):10s} {'lane per point' if per_point else 'lane per float':15s} {s.elapsed_time(e) / 10 * 1e3:9.1f} us")
        os.environ.pop("GENS_K2_SCATTER_PER_POINT", None)
