#!/usr/bin/env python3
"""Micro-benchmark of K1 (volume build) at the benchmark shape: 5 views 480x640, dims 256/128/64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gens_amd import ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
feats = [f.to(dev) for f in sc["features"]]
intrs, c2ws = sc["intrs"].to(dev), sc["c2ws"].to(dev)
w2c = torch.linalg.inv(c2ws).contiguous()
for lvl, d in enumerate([256, 128, 64]):
    tex = ops.pack_nchw(feats[lvl])
    nv, h, w, _ = tex.shape
    k = intrs.clone()
    k[:, :2] *= 0.5 ** lvl                                                    # pre-scaled, as ops.volume_build hands them over
    fn = lambda: ops._VolumeBuild.apply(tex, w2c, k, 1.0, d, 1)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    a = nv * h * w * 16 + 36 * d ** 3
    print(f"K1 D={d}: {ms * 1e3:8.1f} us  {a / 1e6:7.1f} MB algorithmic  {a / ms / 1e6:7.1f} GB/s  ({a / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s)")

# the whole scene (3 levels): one launch (production) against level-by-level launches (GENS_K1_PER_LEVEL=1)
import statistics  # noqa: E402
dims = [256, 128, 64]
a = sum(5 * (480 >> i) * (640 >> i) * 16 + 36 * d ** 3 for i, d in enumerate(dims))
with torch.no_grad():
    for _ in range(100):
        ops.volume_build(feats[:3], intrs, c2ws, dims)
    for env in (None, "GENS_K1_PER_LEVEL"):
        if env:
            os.environ[env] = "1"
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.volume_build(feats[:3], intrs, c2ws, dims)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 20 * 1e3)
        us = statistics.median(ts)
        print(f"K1 scene ({'per level' if env else 'one launch'}, incl. texel packing + matrix inverse): {us:8.1f} us  {a / 1e6:7.1f} MB algorithmic  {a / us / 8e6 * 100:.1f}% of 8 TB/s")
        if env:
            del os.environ[env]
