#!/usr/bin/env python3
"""Per-tap check of ops.conv3d against torch on the CPU: a weight tensor with a single 1 at tap t picks out one shifted copy of the input,
so a wrong tap shows up by index.  usage: conv_tap_debug.py <stride> <x> <y> <z>"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import ops  # noqa: E402

stride, dims = int(sys.argv[1]), tuple(int(a) for a in sys.argv[2:5])
x = torch.randn(1, 2, *dims)
for t in range(27):
    w = torch.zeros(1, 2, 27)
    w[0, 1, t] = 1.0
    w = w.reshape(1, 2, 3, 3, 3)
    ref = F.conv3d(x, w, None, stride=stride, padding=1)
    got = ops.conv3d(x.cuda(), w.cuda(), None, stride).cpu()
    bad = (got - ref).abs() > 1e-6
    if bad.any():
        idx = bad.nonzero()[:4].tolist()
        print(f"tap {t} (dx,dy,dz)=({t // 9},{t // 3 % 3},{t % 3}): {int(bad.sum())} of {bad.numel()} wrong, first {idx}")
print("done")
