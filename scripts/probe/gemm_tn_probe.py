#!/usr/bin/env python3
"""K14 (gens_gemm_tn) against torch.matmul for the weight-gradient shapes of a training step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import ops  # noqa: E402


def t(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for k, m, n in [(61835, 128, 188), (61835, 101, 188), (61835, 128, 27), (247340, 32, 32), (247340, 64, 69), (247340, 33, 32), (247340, 23, 16), (247340, 16, 37), (247340, 8, 16),
                (247340, 33, 1), (61835, 101, 1)]:
    a, b = torch.randn(k, m, device="cuda"), torch.randn(k, n, device="cuda")
    t_mine, t_torch = t(lambda: ops.matmul_tn(a, b)), t(lambda: a.t() @ b)
    fl = 2 * k * m * n
    print(f"{m:4d} x {k:6d} x {n:3d}: gens_gemm_tn {t_mine:7.1f} us ({fl / t_mine / 1e6:6.1f} TFLOP/s, {4 * k * (m + n) / t_mine / 1e3:6.0f} GB/s)   torch {t_torch:7.1f} us ({fl / t_torch / 1e6:6.1f} TFLOP/s)")
