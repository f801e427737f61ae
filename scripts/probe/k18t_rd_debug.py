#!/usr/bin/env python3
"""debug: the operand rows of ray_dir_fc.0 from the two K18 backward kernels, side by side"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from gens_amd import lib as L  # noqa: E402
import test_hip_blend as T  # noqa: E402

nv, n_levels, n = 5, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 16
use_index = len(sys.argv) > 2
ops, net, views, pts = T._setup(nv, n_levels, seed=1, n=n)
dev = pts.device
s, f = nv - 1, 3 + 4 * n_levels
g_rgb = torch.randn(n, 3, generator=torch.Generator().manual_seed(0)).cuda()
w = [p.detach().reshape(-1).contiguous() if p.dim() == 0 else p.detach().contiguous() for p in ops.blend_params(net)]
feats = [ops.aligned16(t.detach()) for t in views.feat_tex]
imgs = ops.aligned16(views.imgs_tex.detach())
hw = [d for t in feats for d in t.shape[1:3]]
idx = torch.randperm(n, generator=torch.Generator().manual_seed(n)).cuda() if use_index else None
n_live = max(1, (3 * n) // 4) if use_index else n
count = torch.tensor([n_live], dtype=torch.int32, device=dev) if use_index else None
args = (L.ptr_table(feats, align=16), L.int_table(hw), n_levels, L.ptr(imgs, align=16), L.ptr(views.w2c), L.ptr(views.intr), L.ptr(views.c2w), nv,
        L.ptr_table(w), L.ptr(pts), L.ptr(idx, torch.int64) if use_index else None, n, L.ptr(count, torch.int32) if use_index else None, L.ptr(g_rgb))
ins = [4, 16, 3 * f, 64, 32, 32, 32, 32, 37, 16, 8]
outs = [16, f, 64, 32, 32, 33, 32, 1, 16, 8, 1]
ev = lambda x: (x + 1) // 2 * 2  # noqa: E731
lib = L.load()
rows_a = lib.gens_blend_train_rows(n, nv)
r_a = [torch.zeros(rows_a, ev(k + 1), device=dev) for k in ins]
l_a = [torch.zeros(rows_a, ev(m), device=dev) for m in outs]
gf, sp_a = torch.zeros(n, s, f, device=dev), torch.zeros(rows_a // 32, device=dev)
L.call("gens_blend_train_bwd", *args, L.ptr_table(r_a), L.ptr_table(l_a), L.ptr(gf), L.ptr(sp_a), L.stream())
rows_b = 16 * (-(-n // 4))
r_b = [torch.full((rows_b, ev(k + 1)), -7.0, device=dev) for k in ins]
l_b = [torch.full((rows_b, ev(m)), -7.0, device=dev) for m in outs]
csz, n_parts = lib.gens_blend_train_acc_floats(n_levels), lib.gens_blend_train_t_parts(n, nv)
sp_b = torch.zeros(n_parts, device=dev)
parts, cc = torch.zeros(n_parts, csz, device=dev), torch.zeros(csz, device=dev)
L.call("gens_blend_train_bwd_t_dump", *args, L.ptr(gf), L.ptr(sp_b), L.ptr(parts), L.ptr(cc), L.ptr_table(r_b), L.ptr_table(l_b), L.stream())
torch.cuda.synchronize()
torch.set_printoptions(precision=4, linewidth=200, sci_mode=False)
pt = torch.arange(n_live, device=dev).repeat_interleave(s)
vw = torch.arange(s, device=dev).repeat(n_live)
row_a = (pt // (32 // s)) * 32 + (pt % (32 // s)) * s + vw
row_b = (pt // 4) * 16 + (pt % 4) * 4 + vw
d = (r_a[0][row_a] - r_b[0][row_b]).abs().max(1)[0]
bad = torch.nonzero(d > 1e-5)[:, 0]
print("R[0]: rows that differ:", int(bad.numel()), "of", int(d.numel()), "first:", bad[:40].tolist())
for k in bad[:6].tolist():
    print("  (pt %d, view %d) row_b %d (tile %d, lane %d):" % (int(pt[k]), int(vw[k]), int(row_b[k]), int(row_b[k]) // 16, int(row_b[k]) % 16), r_a[0][row_a[k]].cpu().tolist(), r_b[0][row_b[k]].cpu().tolist())
print("row-major R[0] rows 0..7:\n", r_a[0][:8].cpu())
print("transposed R[0] rows 0..7:\n", r_b[0][:8].cpu())
print("rgb_fc.0 R cols 32..37, row-major / transposed:\n", r_a[8][:4, 32:38].cpu(), "\n", r_b[8][:4, 32:38].cpu())
off = 0
for l, (m, k) in enumerate(zip(outs, ins)):
    mm, kk = ev(m), ev(k + 1)
    ref = (l_a[l].double().T @ r_a[l].double())[:m, :k + 1]
    got = cc[off:off + mm * kk].view(mm, kk)[:m, :k + 1].double()
    print(l, "block err", float((got - ref).abs().max()), "of", float(ref.abs().max()))
    if l == 0:
        print(ref[:4].cpu(), "\n", got[:4].cpu())
    off += mm * kk
