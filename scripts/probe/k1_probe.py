#!/usr/bin/env python3
"""What bounds K1?  Times the one-round-per-view forward kernel with parts of its memory traffic removed
(scripts/probe/k1_probe.hip) and the library's kernels, on the same preallocated buffers, with warm clocks (the first ~10 ms
after idle run up to 25 % slower) and interleaved repeats (median of 5 blocks of 20 launches)."""
import ctypes as C
import os
import statistics
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from gens_amd import lib as L, ops, synthetic  # noqa: E402

so = os.path.join(HERE, "k1_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
                           os.path.join(HERE, "k1_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.k1_probe.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 2 + [C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
intrs, c2ws = sc["intrs"].to(dev), sc["c2ws"].to(dev)
w2c = torch.linalg.inv(c2ws).contiguous()
names = ["baseline", "all lanes read texel 0", "no texel loads", "no stores", "one tap of four", "no loads, no stores", "non-temporal stores"]


def block(fn, n=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for lvl, d in enumerate([256, 128]):
    tex = ops.pack_nchw(sc["features"][lvl].to(dev))
    nv, h, w, _ = tex.shape
    k = intrs.clone()
    k[:, :2] *= 0.5 ** lvl
    vol, mask = torch.empty(8, d, d, d, device=dev), torch.empty(d, d, d, device=dev)
    cases = []
    for var, name in enumerate(names):
        cases.append((f"probe variant {var} ({name})", None,
                      lambda var=var: lib.k1_probe(tex.data_ptr(), w2c.data_ptr(), k.data_ptr(), nv, h, w, d, vol.data_ptr(), mask.data_ptr(), var,
                                                   torch.cuda.current_stream().cuda_stream)))
    for env in (None, "GENS_K1_SINGLE", "GENS_K1_GENERIC"):
        cases.append((f"libgens_hip {env or 'production'}", env,
                      lambda: L.call("gens_volume_build_fwd", L.ptr(tex), L.ptr(w2c), L.ptr(k), 1.0, nv, h, w, d, 1, L.ptr(vol), L.ptr(mask), L.stream())))
    for _ in range(300):                                    # ~100 ms of work: clocks up
        cases[0][2]()
    torch.cuda.synchronize()
    times = {c[0]: [] for c in cases}
    for _ in range(5):
        for name, env, fn in cases:
            if env:
                os.environ[env] = "1"
            fn()
            times[name].append(block(fn))
            if env:
                del os.environ[env]
    a = nv * h * w * 16 + 36 * d ** 3
    for name, _, _ in cases:
        us = statistics.median(times[name])
        print(f"D={d} {name}: {us:8.1f} us (min {min(times[name]):.1f})  {a / us / 8e6 * 100:5.1f}% of 8 TB/s")
