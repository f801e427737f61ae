#!/usr/bin/env python3
"""Stand-alone K2 backward (gens_lookup_volume_bwd with volume gradients, planar volumes 256 / 128 / 64, 1 M points): how much of its time is the ORDER of
the points?  The same points (a) as drawn (uniform in the cube), (b) sorted by 8^3-voxel brick in Morton order, (c) ray-ordered samples."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
dims = [256, 128, 64]
vols = [torch.randn(1, 4, d, d, d, generator=g).to(dev).requires_grad_(True) for d in dims]
n = 1 << 20


def morton(b):          # (n, 3) int64 brick coordinates < 32 -> 15-bit key
    key = torch.zeros(b.shape[0], dtype=torch.long, device=b.device)
    for bit in range(5):
        for a in range(3):
            key |= ((b[:, a] >> bit) & 1) << (3 * bit + a)
    return key


def timed(pts, label):
    p = pts.clone().requires_grad_(True)
    times = {}
    for it in range(12):
        for v in vols:
            v.grad = None
        f = ops.lookup_volume(p, vols)
        L.profile_begin(only={"gens_lookup_volume_bwd"})
        f.sum().backward()
        rec = L.profile_end(raw=True)
        if it >= 2:
            times.setdefault("bwd", []).append(sum(ms for _, ms, _, _ in rec) * 1e3)
    t = sorted(times["bwd"])
    print(f"{label:46s} gens_lookup_volume_bwd median {t[len(t) // 2]:8.1f} us")


rand = (torch.rand(n, 3, generator=g) * 2 - 1).to(dev)
timed(rand, "uniform random points, as drawn")
brick = ((rand + 1) / 2 * 32).long().clamp(0, 31)
order = torch.argsort(morton(brick))
timed(rand[order].contiguous(), "the same, sorted by brick (Morton order)")
fine = ((rand + 1) / 2 * 255).long().clamp(0, 255)
key = (fine[:, 0] * 256 + fine[:, 1]) * 256 + fine[:, 2]
timed(rand[torch.argsort(key)].contiguous(), "the same, sorted by voxel (x, y, z)")
# ray-ordered samples: 8192 rays x 128 samples through the cube
o = (torch.rand(8192, 1, 3, generator=g) * 0.6 - 0.3).to(dev)
d = torch.nn.functional.normalize(torch.randn(8192, 1, 3, generator=g), dim=-1).to(dev)
tt = torch.linspace(-0.9, 0.9, 128, device=dev).view(1, 128, 1)
timed((o + d * tt).reshape(-1, 3).contiguous(), "ray-ordered samples (8 192 rays x 128)")
# what a sort costs (torch.sort on the device, 1 M keys)
for dt in (torch.int32, torch.int64):
    k = key.to(dt)
    for _ in range(3):
        torch.sort(k)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        torch.sort(k)
    e.record()
    torch.cuda.synchronize()
    print(f"torch.sort of {n} {dt} keys: {s.elapsed_time(e) * 100:.1f} us")
