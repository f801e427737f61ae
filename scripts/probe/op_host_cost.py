"""Host time per call (enqueue only, queue never drained inside the loop) of the 2-D trunk's operators on a deep-stage map: K22 / K21 / K16 against
the aten operators they replace."""
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops  # noqa: E402
from gens_amd.models.modules import feature_network as fn  # noqa: E402

L.load()
dev = torch.device("cuda:0")
x = torch.randn(5, 96, 30, 40, device=dev, requires_grad=True)


class _Id(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        return a.view_as(a)

    @staticmethod
    def backward(ctx, ga):
        return ga


def host_us(f, n=300):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


bn = fn.BatchNorm2dReLU(96, relu=True).to(dev).train()
ref = nn.BatchNorm2d(96).to(dev).train()
dw = fn.DepthwiseConv2d(96, 96, 3, padding=1, groups=96, bias=False).to(dev)
dwr = nn.Conv2d(96, 96, 3, padding=1, groups=96, bias=False).to(dev)
g = torch.randn_like(x)
print(f"K22 BatchNorm + ReLU forward            {host_us(lambda: bn(x)):7.1f} us host per call")
print(f"aten BatchNorm + ReLU forward           {host_us(lambda: F.relu(ref(x))):7.1f}")
print(f"K22 forward + backward                  {host_us(lambda: bn(x).backward(g)):7.1f}")
print(f"aten forward + backward                 {host_us(lambda: F.relu(ref(x)).backward(g)):7.1f}")
print(f"K21 depth-wise 3 x 3 forward            {host_us(lambda: dw(x)):7.1f}")
print(f"aten depth-wise forward                 {host_us(lambda: dwr(x)):7.1f}")
print(f"K21 forward + backward                  {host_us(lambda: dw(x).backward(g)):7.1f}")
print(f"aten forward + backward                 {host_us(lambda: dwr(x).backward(g)):7.1f}")
print(f"K16 instance norm + ReLU forward        {host_us(lambda: ops.instnorm_relu(x)):7.1f}")
print(f"aten instance norm + ReLU forward       {host_us(lambda: F.relu(F.instance_norm(x))):7.1f}")
print(f"K16 forward + backward                  {host_us(lambda: ops.instnorm_relu(x).backward(g)):7.1f}")
print(f"aten forward + backward                 {host_us(lambda: F.relu(F.instance_norm(x)).backward(g)):7.1f}")
print(f"empty autograd.Function (apply only)    {host_us(lambda: _Id.apply(x)):7.1f}")
print(f"torch.empty_like                        {host_us(lambda: torch.empty_like(x)):7.1f}")
