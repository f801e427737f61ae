#!/usr/bin/env python3
"""Workload for the PMC traffic passes of K15 / K16 (scripts/pmc_traffic.py): ONE layer shape, so that per-kernel averages are per-shape
figures -- the U-Net's first block at 256^3, conv 8 -> 8 (stride 1) + instance-norm + ReLU, forward and backward, twice."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(1, 8, 256, 256, 256, device=dev, generator=g).requires_grad_(True)
w = (torch.randn(8, 8, 3, 3, 3, device=dev, generator=g) / 216 ** 0.5).requires_grad_(True)
cot = torch.randn(1, 8, 256, 256, 256, device=dev, generator=g)
for _ in range(2):
    y = ops.instnorm_relu(ops.conv3d(x, w, None, 1))
    torch.autograd.grad(y, [x, w], cot)
torch.cuda.synchronize()
print("done")
