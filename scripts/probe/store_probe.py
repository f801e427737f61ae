#!/usr/bin/env python3
"""Store side of K1 alone: how fast can 9 planes of 256^3 floats (604 MB) be written, and with which store width?"""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "store_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", os.path.join(HERE, "store_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.store_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
d = 256
vol, mask = torch.empty(8, d, d, d, device=dev), torch.empty(d, d, d, device=dev)
nbytes = 9 * d ** 3 * 4


def timeit(fn, name):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"{name}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s")


for var, name in enumerate(["dword stores", "dwordx4 stores", "dword stores + 400 FMAs", "dwordx2 stores (variant 3 is the switch default)", "400 FMAs + dword stores, 4 chunks per wave",
                            "400 FMAs + dword stores, 8 chunks per wave", "400 FMAs + dword stores, 16 chunks per wave",
                            "400 FMAs + dword stores, 32 chunks per wave"]):
    timeit(lambda: lib.store_probe(vol.data_ptr(), mask.data_ptr(), d, var, torch.cuda.current_stream().cuda_stream), name)
both = torch.empty(9, d, d, d, device=dev)
timeit(lambda: both.fill_(1.0), "torch fill_ (9 planes)")
timeit(lambda: both.zero_(), "torch zero_ (9 planes)")
