#!/usr/bin/env python3
"""The gather ceiling the lookup kernels run against: 16-byte texel loads per second by table size (L1 / L2 / Infinity Cache / HBM resident) and by
how many adjacent lanes share a 128-byte line.  python scripts/probe/gather_rate_probe.py"""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "gather_rate_probe.so")
src = os.path.join(HERE, "gather_rate_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", so])
lib = C.CDLL(so)
lib.gather_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
table = torch.randn(1 << 26, 4, device=dev)                    # 1 GiB of texels
blocks, per_thread = 16384, 64
out = torch.empty(blocks * 256, 4, device=dev)
loads = blocks * 256 * per_thread
print(f"{loads / 1e6:.0f} M 16-byte loads per launch, {blocks} workgroups of 256")
print(f"{'table':>10s} {'lanes/line':>10s} {'us':>9s} {'G loads/s':>10s} {'G lines/s':>10s} {'loads/clk/CU @2.4GHz':>22s}")
for size_log, name in ((14, "16 KiB"), (17, "128 KiB"), (21, "2 MiB"), (25, "32 MiB"), (27, "128 MiB"), (30, "1 GiB")):
    for share_log in (0, 1, 2, 3):
        n_lines = 1 << (size_log - 7)
        fn = lambda: lib.gather_probe(table.data_ptr(), n_lines, per_thread, share_log, blocks, out.data_ptr(), torch.cuda.current_stream().cuda_stream)  # noqa: E731
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 5 * 1e3
        g = loads / us / 1e3
        print(f"{name:>10s} {1 << share_log:10d} {us:9.1f} {g:10.1f} {g / (1 << share_log):10.1f} {g / 256 / 2.4:22.2f}")
