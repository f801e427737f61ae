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
