#!/usr/bin/env python3
"""K17's backward launch and the weight-gradient products behind it, timed by HIP events inside the entry points (62 000 points, three levels).
With GENS_HIP_LIB=<a build with -DGENS_K17_NO_RH> the same without the operand rows `rh` being written: the most that forming them inside K14 could save."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L  # noqa: E402
from gens_amd import ops  # noqa: E402
from oracle import sdf_train_oracle as T  # noqa: E402   (weights of the shipped shape only)

n_levels = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 62000
g = torch.Generator().manual_seed(0)
dims = [256, 128, 64, 32, 16][:n_levels]
vols = [(0.5 * torch.randn(1, 4, d, d, d, generator=g)).cuda().requires_grad_(True) for d in dims]
W, b = T.shipped_weights(n_levels, seed=1, scale=1.0)
W = [w.cuda().requires_grad_(True) for w in W]
b = [v.cuda().requires_grad_(True) for v in b]
pts = (torch.rand(n, 3, generator=g) * 2 - 1).cuda()
step = ops.SdfTrainStep(W, b, vols, ops.VolumeSet.packed(vols))
times = {}
for it in range(25):
    L.profile_begin()
    y, gr, s = step(pts)
    (y.sum() + gr.sum() + s.sum()).backward()
    rec = L.profile_end(raw=True)
    if it >= 5:
        for name, ms, _, _ in rec:
            times.setdefault(name, []).append(ms)
for name, t in sorted(times.items(), key=lambda kv: -sum(kv[1])):
    per = len(t) // 20
    t2 = sorted(t)
    print(f"{name:36s} {per} per step, median {t2[len(t2) // 2] * 1e3:8.1f} us   sum per step {sum(t) / 20 * 1e3:8.1f} us")
