#!/usr/bin/env python3
"""Where one wave of K18t's backward spends its cycles: stamps at the phase boundaries of the fourth tile of workgroup 7's waves.
Build:  make -C gens_amd/csrc stamps      Run:  GENS_HIP_LIB=gens_amd/csrc/stamps/libgens_hip_k18t_stamps.so python scripts/probe/k18t_stamps_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = sys.argv[:1] + (sys.argv[1:] or ["62000", "5", "5"])
from scripts.probe import k18t_probe  # noqa: E402
from gens_amd import lib as L  # noqa: E402

k18t_probe.main()
buf = (ctypes.c_ulonglong * (4 * 64))()
assert L.load().gens_debug_k18t_stamps(buf) == 0
names = ["points (prefetched), projection, texel loads issued, compute_angle", "ray_dir_fc", "texels interpolated, view weights, mean / variance", "base_fc",
         "vis_fc, vis_fc2, rgb_fc, soft-max", "reverse rgb_fc, vis_fc2", "reverse vis_fc", "dW base_fc.2, reverse base_fc.2, dW base_fc.0",
         "reverse base_fc.0, mean / var, ray_dir_fc", "g_feat stores"]
for w in range(4):
    st = [buf[w * 64 + i] for i in range(64)]
    st = [v for v in st if v]
    if len(st) < 2:
        print(f"wave {w}: no stamps")
        continue
    d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
    print(f"wave {w}: {len(st)} stamps, tile total {st[-1] - st[0]} shader cycles (s_memtime)")
    for i, v in enumerate(d):
        print(f"   {names[i] if i < len(names) else '?':48s} {v:8d}")
