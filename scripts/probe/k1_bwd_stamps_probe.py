#!/usr/bin/env python3
"""Where the workgroups of K1's backward (volume_bwd_tiles_k) spend their cycles: wave 0's clock per phase, summed over all workgroups of a launch.
Build:  make -C gens_amd/csrc stamps      Run:  GENS_HIP_LIB=gens_amd/csrc/stamps/libgens_hip_k1b_stamps.so python scripts/probe/k1_bwd_stamps_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
nv, h0, w0 = (3, 1152, 1600) if "--conf-shape" in sys.argv else (5, 480, 640)
dims = [256, 128, 64, 32, 16] if "--levels5" in sys.argv or "--conf-shape" in sys.argv else [256, 128, 64]
n = len(dims)
sc = synthetic.make_scene(nv=nv, h=h0, w=w0, n_levels=5, seed=0)
cams = ops.SceneCams.of(sc["intrs"].to(dev), sc["c2ws"].to(dev))
texs = [ops.pack_nchw(sc["features"][l].to(dev)) for l in range(n)]
hw = [x for t in texs for x in (t.shape[1], t.shape[2])]
vols = [torch.empty(8, d, d, d, device=dev) for d in dims]
masks = [torch.empty(d, d, d, device=dev) for d in dims]
counts = [torch.empty(d ** 3, device=dev, dtype=torch.uint8) for d in dims]
intrs = [cams.ks[l] for l in range(n)]
L.call("gens_volume_build_levels", L.ptr_table(texs), L.int_table(hw), L.int_table(dims), n, L.ptr(cams.w2c), L.ptr_table(intrs), nv, 1,
       L.ptr_table(vols), L.ptr_table(masks), L.ptr_table(counts, torch.uint8), L.stream())
g = torch.Generator(device=dev).manual_seed(1)
gvols = [torch.randn(8, d, d, d, device=dev, generator=g) for d in dims]
need = L.load().gens_volume_build_bwd_levels_scratch_bytes(L.int_table(hw), L.int_table(dims), n, nv)
scratch = torch.empty(need, device=dev, dtype=torch.uint8)
out = [torch.zeros_like(t) for t in texs]


def levels():
    L.call("gens_volume_build_bwd_levels", L.ptr_table(texs), L.int_table(hw), L.int_table(dims), n, L.ptr(cams.w2c), L.ptr_table(intrs), nv, L.ptr_table(vols),
           L.ptr_table(counts, torch.uint8), L.ptr_table(gvols), L.ptr_table(out), L.ptr(scratch, torch.uint8), need, L.stream())


for _ in range(3):
    levels()
torch.cuda.synchronize()
fn = L.load().gens_debug_k1_bwd_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
assert fn(buf, 1) == 0
reps = 10
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    levels()
e.record()
torch.cuda.synchronize()
assert fn(buf, 0) == 0
names = ["take an item", "zero the window", "the item's pairs", "flush the window", "direct items", "(window items)"]
tot = sum(buf[i] for i in range(5))
print(f"{s.elapsed_time(e) / reps:.3f} ms per call (five launches); window items per call {buf[5] / reps:.0f}")
for i in range(5):
    print(f"   {names[i]:20s} {buf[i] / reps / 1e3:12.1f} k cycles per call over all workgroups   {100.0 * buf[i] / max(tot, 1):5.1f} %")
print(f"   per workgroup (256): {tot / reps / 256 / 1e3:.1f} k cycles")
