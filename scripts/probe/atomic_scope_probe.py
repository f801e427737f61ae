#!/usr/bin/env python3
"""Float atomic-add throughput into gradient maps: device scope on one copy against workgroup scope on a private copy per XCD (and whether
blockIdx % 8 is the XCD).  python scripts/probe/atomic_scope_probe.py"""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "atomic_scope_probe.so")
src = os.path.join(HERE, "atomic_scope_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", so])
lib = C.CDLL(so)
lib.atomic_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
n_texels = 5 * (480 * 640 + 240 * 320 + 120 * 160)          # the feature pyramid of five views: 2.0 M float4 texels = 32 MB
maps = torch.zeros(8, n_texels, 4, device=dev)
mismatch = torch.zeros(1, dtype=torch.int32, device=dev)
blocks, per_thread = 8192, 16
total = blocks * 256 * per_thread * 4


def timeit(mode, name):
    fn = lambda: lib.atomic_probe(maps.data_ptr(), n_texels, per_thread, blocks, mode, mismatch.data_ptr(), torch.cuda.current_stream().cuda_stream)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    print(f"{name:58s} {us:9.1f} us  {total / us / 1e3:7.2f} G atomics/s")


timeit(0, "device scope, one copy")
timeit(2, "device scope, one copy per XCD")
timeit(1, "workgroup scope, one copy per XCD")
timeit(3, "workgroup scope, one copy (not coherent across XCDs)")
timeit(4, "device scope, one copy, uint32 adds")
timeit(5, "device scope, one copy, uint64 adds")
timeit(6, "float, 64 lanes on 64 consecutive floats")
timeit(7, "float, 16 lanes on 16 consecutive floats")
print("workgroups whose XCC_ID differs from blockIdx % 8:", int(mismatch.item()))
maps.zero_()
lib.atomic_probe(maps.data_ptr(), n_texels, per_thread, blocks, 1, mismatch.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("sum over the per-XCD copies after one workgroup-scope pass:", float(maps.double().sum()), "expected", float(total))
