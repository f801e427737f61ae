#!/usr/bin/env python3
"""Does K1's time depend on where its output lives?  Same launch into (a) one preallocated buffer, (b) a fresh torch.empty per call,
(c) through ops._VolumeBuild (fresh outputs + autograd bookkeeping)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
intrs, c2ws = sc["intrs"].to(dev), sc["c2ws"].to(dev)
w2c = torch.linalg.inv(c2ws).contiguous()
d = 256
tex = ops.pack_nchw(sc["features"][0].to(dev))
nv, h, w, _ = tex.shape
k = intrs.clone()
vol, mask = torch.empty(8, d, d, d, device=dev), torch.empty(d, d, d, device=dev)


def launch(v, m):
    L.call("gens_volume_build_fwd", L.ptr(tex), L.ptr(w2c), L.ptr(k), 1.0, nv, h, w, d, 1, L.ptr(v), L.ptr(m), L.stream())


def a():
    launch(vol, mask)


def b():
    launch(torch.empty(8, d, d, d, device=dev), torch.empty(d, d, d, device=dev))


def c():
    ops._VolumeBuild.apply(tex, w2c, k, 1.0, d, 1)


keep = []


def e():
    v, m = torch.empty(8, d, d, d, device=dev), torch.empty(d, d, d, device=dev)
    keep.append((v, m))
    if len(keep) > 4:
        keep.pop(0)
    launch(v, m)


for name, fn in [("preallocated", a), ("fresh torch.empty per call", b), ("ops._VolumeBuild.apply", c), ("rotating over 5 buffer pairs", e), ("preallocated again", a)]:
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    t.record()
    torch.cuda.synchronize()
    print(f"{name}: {s.elapsed_time(t) / 20 * 1e3:8.1f} us")
