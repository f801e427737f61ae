#!/usr/bin/env python3
"""Where one workgroup of K17's backward (sdf_train_bwd_k) spends its cycles: stamps at every phase boundary of workgroup 300, all four waves.
Build:  make -C gens_amd/csrc stamps      Run:  GENS_HIP_LIB=gens_amd/csrc/stamps/libgens_hip_stamps.so python scripts/probe/k17_stamps_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L  # noqa: E402
from gens_amd import ops  # noqa: E402
from oracle import sdf_train_oracle as T  # noqa: E402   (weights of the shipped shape only)

n_levels = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 68000
g = torch.Generator().manual_seed(0)
dims = [256, 128, 64, 32, 16][:n_levels]
vols = [(0.5 * torch.randn(1, 4, d, d, d, generator=g)).cuda().requires_grad_(True) for d in dims]
W, b = T.shipped_weights(n_levels, seed=1, scale=1.0)
W = [w.cuda().requires_grad_(True) for w in W]
b = [v.cuda().requires_grad_(True) for v in b]
pts = (torch.rand(n, 3, generator=g) * 2 - 1).cuda()
step = ops.SdfTrainStep(W, b, vols, ops.VolumeSet.packed(vols))
for it in range(3):
    y, gr, s = step(pts)
    (y.sum() + gr.sum() + s.sum()).backward()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4 * 128))()
fn = L.load().gens_debug_k17_stamps
assert fn(buf) == 0
names = ["prologue"]
for l in range(6):
    names += [f"f{l} zero", f"f{l} product", f"f{l} barrier", f"f{l} element-wise"]
names += ["f5 barrier"]
for l in range(5, -1, -1):
    names += [f"r{l} parked loads", f"r{l} product", f"r{l} barrier", f"r{l} element-wise + stores"]
names += ["r0 barrier"]
for w in range(4):
    st = [buf[w * 128 + i] for i in range(128)]
    st = [v for v in st if v]
    d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
    print(f"wave {w}: {len(st)} stamps, total {st[-1] - st[0]} cycles")
    if w == 0:
        for i, v in enumerate(d):
            print(f"   {names[i + 1] if i + 1 < len(names) else '?':32s} {v:8d}")
