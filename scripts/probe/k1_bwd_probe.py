#!/usr/bin/env python3
"""K1 backward: image-tile kernel and LDS-window scatter against direct device atomics (GENS_K1_BWD_DIRECT=1): time and agreement."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
intrs, c2ws = sc["intrs"].to(dev), sc["c2ws"].to(dev)
w2c = torch.linalg.inv(c2ws).contiguous()
for lvl, d in enumerate([256, 128, 64]):
    tex = ops.pack_nchw(sc["features"][lvl].to(dev))
    nv, h, w, _ = tex.shape
    gvol = torch.randn(8, d, d, d, device=dev)
    res = {}
    need = L.load().gens_volume_build_bwd_scratch_bytes(nv, h, w, d)
    scratch = torch.empty(need, device=dev, dtype=torch.uint8)
    for mode in ("tiled", "window", "direct"):
        if mode == "direct":
            os.environ["GENS_K1_BWD_DIRECT"] = "1"
        g = torch.zeros_like(tex)
        if mode == "tiled":
            fn = lambda: L.call("gens_volume_build_bwd_tiled", L.ptr(tex), L.ptr(w2c), L.ptr(intrs), 0.5 ** lvl, nv, h, w, d, L.ptr(gvol), L.ptr(g),  # noqa: E731
                                L.ptr(scratch, torch.uint8), need, L.stream())
        else:
            fn = lambda: L.call("gens_volume_build_bwd", L.ptr(tex), L.ptr(w2c), L.ptr(intrs), 0.5 ** lvl, nv, h, w, d, L.ptr(gvol), L.ptr(g), L.stream())  # noqa: E731
        fn()
        torch.cuda.synchronize()
        res[mode] = g.clone()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        print(f"D={d} {mode}: {s.elapsed_time(e) / 10:8.3f} ms")
        os.environ.pop("GENS_K1_BWD_DIRECT", None)
    for mode in ("tiled", "window"):
        diff = (res[mode] - res["direct"]).abs().max().item()
        print(f"D={d} max |{mode} - direct| = {diff:.3e}  (max |g| = {res['direct'].abs().max().item():.3e})")
    print(f"D={d} scratch of the tiled kernel: {need / 2**20:.0f} MiB")
