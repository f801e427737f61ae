#!/usr/bin/env python3
"""A few launches of the split-half SDF value kernel (gens_sdf_value_f16) on lattice-like points, for counter passes:
python scripts/probe/sdf_value_workload.py [points] [levels] [random|lattice]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import ops, synthetic  # noqa: E402
from gens_amd.config import gens_model_conf  # noqa: E402
from gens_amd.models.modules.implicit_surface import ImplicitSurface  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 1024 * 1024
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else "lattice"
precision = sys.argv[4] if len(sys.argv) > 4 else "f16x2"
want_grad = len(sys.argv) > 5 and sys.argv[5] == "grad"
dims = [256, 128, 64, 32, 16][:levels]
dev = torch.device("cuda:0")
torch.manual_seed(0)
surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"]).to(dev).eval()
vols = ops.VolumeSet.packed([v.to(dev) for v in synthetic.make_volumes(dims, seed=1)])
if mode == "random":
    pts = torch.rand(n, 3, device=dev) * 1.6 - 0.8
else:                                   # 64^3 chunks of a 512^3 lattice, as extract_fields walks them
    side = 64
    g = torch.linspace(-1, 1, 512, device=dev)
    chunks = []
    k = 0
    while sum(c.shape[0] for c in chunks) < n:
        i, j, l = (k * 3) % 8, (k * 5) % 8, k % 8
        xs, ys, zs = g[i * side:(i + 1) * side], g[j * side:(j + 1) * side], g[l * side:(l + 1) * side]
        chunks.append(torch.stack(torch.meshgrid(xs, ys, zs, indexing="ij"), -1).reshape(-1, 3))
        k += 1
    pts = torch.cat(chunks)[:n].contiguous()
plan = ops.SdfMlpPlan(surf.sdf_network)
sdf = torch.empty(n, 1, device=dev)
grad = torch.empty(n, 3, device=dev) if want_grad else None
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops.sdf_mlp(plan, vols, pts, sdf_out=sdf, grad_out=grad, want_grad=want_grad, precision=precision)
torch.cuda.synchronize()
s.record()
for _ in range(5):
    ops.sdf_mlp(plan, vols, pts, sdf_out=sdf, grad_out=grad, want_grad=want_grad, precision=precision)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
tag = precision + (" grad" if want_grad else "")
print(f"{tag} L={levels} {mode}: {ms:.3f} ms per {n} points = {n / ms / 1e3:.1f} Mpts/s")
