#!/usr/bin/env python3
"""K18 backward at the training step's shape: gens_blend_train_bwd_t (transposed, round 6) against gens_blend_train_bwd_acc (32-row workgroups), launch by
launch with HIP events.  python scripts/probe/k18t_probe.py [n_points] [nv] [n_levels]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402
from gens_amd.models.modules.blending_network import BlendingNetwork  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 62000
    nv = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    nl = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(nv=nv, h=480, w=640, n_levels=nl, seed=0)
    torch.manual_seed(0)
    net = BlendingNetwork(d_feature=4 * nl).to(dev)
    views = ops.SceneViews(sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev), [f.to(dev) for f in sc["features"]])
    pts = ((torch.rand(n, 3) * 2 - 1) * 0.6).to(dev)
    g_rgb = torch.randn(n, 3, device=dev)
    s, f = nv - 1, 3 + 4 * nl
    w = [p.detach().reshape(-1).contiguous() if p.dim() == 0 else p.detach().contiguous() for p in ops.blend_params(net)]
    feats = [ops.aligned16(t.detach()) for t in views.feat_tex]
    imgs = ops.aligned16(views.imgs_tex.detach())
    hw = [d for t in feats for d in t.shape[1:3]]
    args = (L.ptr_table(feats, align=16), L.int_table(hw), nl, L.ptr(imgs, align=16), L.ptr(views.w2c), L.ptr(views.intr), L.ptr(views.c2w), nv,
            L.ptr_table(w), L.ptr(pts), None, n, None, L.ptr(g_rgb))
    lib = L.load()
    csz = lib.gens_blend_train_acc_floats(nl)
    ins = [4, 16, 3 * f, 64, 32, 32, 32, 32, 37, 16, 8]
    outs = [16, f, 64, 32, 32, 33, 32, 1, 16, 8, 1]
    macs = sum(a * b for a, b in zip(ins, outs))
    flops = n * s * (2 * 2 * macs + 2 * sum(m * (k + 1) for m, k in zip(outs, ins)))
    gf = torch.zeros(n, s, f, device=dev)
    res = {}
    for name, entry, n_parts, n_s in (("rowmajor_acc", "gens_blend_train_bwd_acc", lib.gens_blend_train_acc_parts(n, nv), lib.gens_blend_train_rows(n, nv) // 32),
                                      ("transposed", "gens_blend_train_bwd_t", lib.gens_blend_train_t_parts(n, nv), lib.gens_blend_train_t_parts(n, nv))):
        parts, cc, sp = torch.zeros(n_parts, csz, device=dev), torch.zeros(csz, device=dev), torch.zeros(n_s, device=dev)
        call = lambda: L.call(entry, *args, L.ptr(gf), L.ptr(sp), L.ptr(parts), L.ptr(cc), L.stream())  # noqa: E731
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            call()
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        res[name] = (ts[len(ts) // 2], cc.clone(), float(sp.sum()), gf.clone())
        print(f"{name:14s} median {ts[len(ts) // 2] * 1e3:8.1f} us  p10 {ts[3] * 1e3:8.1f}  parts {n_parts}  -> {flops / ts[len(ts) // 2] / 1e9:6.1f} TFLOP/s "
              f"(forward again + reverse + [dW | db]: {flops / 1e9:.2f} GFLOP)")
    a, b = res["rowmajor_acc"], res["transposed"]
    off = 0
    worst = 0.0
    for m, k in zip(outs, ins):
        mm, kk = (m + 1) // 2 * 2, (k + 2) // 2 * 2
        x, y = a[1][off:off + mm * kk].view(mm, kk)[:m, :k + 1], b[1][off:off + mm * kk].view(mm, kk)[:m, :k + 1]
        worst = max(worst, float((x - y).abs().max()) / max(float(x.abs().max()), 1e-9))
        off += mm * kk
    print(f"blocks: worst relative difference {worst:.1e}; s: {a[2]:.6e} / {b[2]:.6e}; g_feat: {float((a[3] - b[3]).abs().max()):.1e} of {float(a[3].abs().max()):.1e}")


if __name__ == "__main__":
    main()
