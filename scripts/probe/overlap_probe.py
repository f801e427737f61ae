#!/usr/bin/env python3
"""Do VALU work and stores overlap?  (scripts/probe/overlap_probe.hip)"""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "overlap_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", os.path.join(HERE, "overlap_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.overlap_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
d = 256
vol = torch.empty(9, d, d, d, device=dev)
for nf, ns, wrap in [(0, 9, 0), (400, 1, 0), (400, 9, 0), (400, 9, 1), (0, 9, 1), (200, 9, 0), (800, 9, 0), (800, 1, 0), (400, 3, 0), (400, 5, 0)]:
    fn = lambda: lib.overlap_probe(vol.data_ptr(), d, nf, ns, wrap, torch.cuda.current_stream().cuda_stream)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    print(f"{nf:4d} FMAs + {ns} stores{' (1 MiB window)' if wrap else '':16s}: {s.elapsed_time(e) / 20 * 1e3:8.1f} us")
