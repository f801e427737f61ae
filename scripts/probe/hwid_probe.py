#!/usr/bin/env python3
"""Bit fields of HW_REG_HW_ID / HW_REG_XCC_ID over the waves of a launch: python scripts/probe/hwid_probe.py"""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so, src = os.path.join(HERE, "hwid_probe.so"), os.path.join(HERE, "hwid_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", so])
lib = C.CDLL(so)
lib.hwid_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
n = 1 << 16
out = torch.zeros(2 * n, device="cuda:0", dtype=torch.int32)
lib.hwid_probe(out.data_ptr(), n, 50, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
a = out.cpu().numpy().view(np.uint32).reshape(n, 2)
hw, xcc = a[:, 0], a[:, 1]
print("distinct XCC_ID values (low 8 bits):", sorted(set((xcc & 0xFF).tolist())), " raw or:", hex(int(np.bitwise_or.reduce(xcc))))
print("HW_ID bits ever set:", hex(int(np.bitwise_or.reduce(hw))), " bits always set:", hex(int(np.bitwise_and.reduce(hw))))
for name, lo, w in (("wave_id", 0, 4), ("simd_id", 4, 2), ("pipe_id", 6, 2), ("cu_id", 8, 4), ("sh_id", 12, 1), ("se_id", 13, 3), ("tg_id", 16, 4)):
    v = (hw >> lo) & ((1 << w) - 1)
    print(f"{name:8s} bits [{lo + w - 1}:{lo}]: values {sorted(set(v.tolist()))}")
key = ((xcc & 0xF).astype(np.int64) << 16) | ((hw >> 4) & 0xFFF)
print("distinct (xcc, se, sh, cu, pipe, simd):", len(set(key.tolist())), " distinct (xcc, se, sh, cu):", len(set((key >> 4).tolist())))
