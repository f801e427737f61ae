#!/usr/bin/env python3
"""gens_sdf_grad_f16 (split-half value + gradient) against gens_sdf_grad (float32) on the headline shape: errors and time per launch.
    python scripts/probe/k6gh_probe.py [--n 3400000] [--dims 256 128 64] [--reps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=3400000)
    ap.add_argument("--dims", type=int, nargs="+", default=[256, 128, 64])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--small", action="store_true", help="the tests' tiny volumes (16, 12, 8)")
    ap.add_argument("--value", action="store_true", help="time the value-only kernels too (gens_sdf_value / gens_sdf_value_f16)")
    args = ap.parse_args()
    from gens_amd import ops, synthetic
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.sdf_network import SDFNetwork
    dims = (16, 12, 8) if args.small else tuple(args.dims)
    torch.manual_seed(3)
    net = SDFNetwork(**gens_model_conf(volume_dims=dims)["implicit_surface"]["sdf_network"])
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p) * (p.abs().mean() + 0.02))
    net = net.cuda()
    vols = ops.VolumeSet.packed([v.cuda() for v in synthetic.make_volumes(dims, seed=9)])
    pts = (torch.rand(args.n, 3, generator=torch.Generator().manual_seed(1)) * 2.1 - 1.05).cuda()
    plan = ops.SdfMlpPlan(net)
    print("pieces", None if plan.grad_pieces is None else tuple(plan.grad_pieces.shape), "g_scale", getattr(plan, "grad_scale", None))
    s32, g32 = ops.sdf_mlp(plan, vols, pts, want_grad=True)
    s16, g16 = ops.sdf_mlp(plan, vols, pts, want_grad=True, precision="f16x2")
    torch.cuda.synchronize()
    print("overflowed", plan.overflowed())
    ds, dg = (s16 - s32).abs(), (g16 - g32).abs()
    print("sdf  max abs err %.3e  mean %.3e  (|sdf| max %.3f)" % (ds.max(), ds.mean(), s32.abs().max()))
    print("grad max abs err %.3e  mean %.3e  (|grad| max %.3f mean %.3f)" % (dg.max(), dg.mean(), g32.abs().max(), g32.abs().mean()))
    bad = (dg.max(1).values > 1e-3).nonzero().flatten()
    print("rows with a gradient error above 1e-3:", bad.numel(), bad[:10].tolist())
    for prec in ("f32", "f16x2"):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.reps + 1)]
        ops.sdf_mlp(plan, vols, pts, want_grad=True, precision=prec)
        ev[0].record()
        for i in range(args.reps):
            ops.sdf_mlp(plan, vols, pts, want_grad=True, precision=prec)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.reps))
        print("%-6s %.3f ms per launch of %d points (median of %d; min %.3f)" % (prec, ms[len(ms) // 2], args.n, args.reps, ms[0]))
    if args.value:
        v32 = ops.sdf_mlp(plan, vols, pts)
        v16 = ops.sdf_mlp(plan, vols, pts, precision="f16x2")
        print("value-only: max abs err %.3e, overflowed %s" % ((v16 - v32).abs().max(), plan.overflowed()))
        for prec in ("f32", "f16x2"):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.reps + 1)]
            ops.sdf_mlp(plan, vols, pts, precision=prec)
            ev[0].record()
            for i in range(args.reps):
                ops.sdf_mlp(plan, vols, pts, precision=prec)
                ev[i + 1].record()
            torch.cuda.synchronize()
            ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.reps))
            print("value %-6s %.3f ms per launch of %d points (median of %d; min %.3f)" % (prec, ms[len(ms) // 2], args.n, args.reps, ms[0]))


if __name__ == "__main__":
    main()
