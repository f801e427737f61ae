#!/usr/bin/env python3
"""LDS atomic-add cost per wave instruction by operand type and address pattern (K1 backward's image-tile kernel accumulates in LDS).
python scripts/probe/lds_atomic_probe.py"""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "lds_atomic_probe.so")
src = os.path.join(HERE, "lds_atomic_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", so])
lib = C.CDLL(so)
lib.lds_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
out = torch.zeros(4, device="cuda:0")
blocks, iters = 256 * 8, 4096              # 8 workgroups of 4 waves per CU
clock_ghz = 2.4
for tname, t in (("f32 atomic", 0), ("u32 atomic", 1), ("u64 atomic", 2), ("f32 load+add+store", 3), ("f64 atomic", 4), ("f32 native (wg scope)", 5)):
    for pname, p in (("64 distinct words", 0), ("one word", 1), ("runs of 4 lanes", 2), ("runs of 16 lanes", 3), ("distinct, stride 4", 4)):
        fn = lambda: lib.lds_probe(t, p, iters, blocks, out.data_ptr(), torch.cuda.current_stream().cuda_stream)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 3 * 1e3
        per_cu = blocks * 4 * iters / 256                       # wave instructions per CU
        print(f"{tname:20s} {pname:20s} {us:9.1f} us   {us * 1e3 * clock_ghz / per_cu:7.1f} cycles per wave instruction per CU")
