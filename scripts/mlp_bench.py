#!/usr/bin/env python3
"""Micro-benchmark of the fused MLP kernels (development aid): points/s and TFLOP/s of gens_sdf_mlp / gens_blend_views."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402
from gens_amd.config import gens_model_conf  # noqa: E402
from gens_amd.models.modules.implicit_surface import ImplicitSurface  # noqa: E402


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    dev = torch.device("cuda:0")
    for dims in ([256, 128, 64], [256, 128, 64, 32, 16]):
        torch.manual_seed(0)
        surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"]).to(dev).eval()
        vols = ops.VolumeSet.packed([v.to(dev) for v in synthetic.make_volumes(dims, seed=1)])
        pts = (torch.rand(n, 3, device=dev) * 1.6 - 0.8)
        plan = ops.SdfMlpPlan(surf.sdf_network)
        fe = 20 * len(dims)
        f_fwd = 2 * (27 * 128 + (128 + fe) * (4 * 128 + 101 + 1))
        sdf = torch.empty(n, 1, device=dev)
        grad = torch.empty(n, 3, device=dev)
        t1 = timeit(lambda: ops.sdf_mlp(plan, vols, pts, sdf_out=sdf))
        t2 = timeit(lambda: ops.sdf_mlp(plan, vols, pts, want_grad=True, sdf_out=sdf, grad_out=grad))
        print(f"L={len(dims)} sdf fwd : {t1:7.2f} ms  {n / t1 / 1e3:7.1f} Mpts/s  {n * f_fwd / t1 / 1e9:6.1f} TFLOP/s")
        print(f"L={len(dims)} sdf grad: {t2:7.2f} ms  {n / t2 / 1e3:7.1f} Mpts/s  {n * 2 * f_fwd / t2 / 1e9:6.1f} TFLOP/s")
        s32, g32 = sdf.clone(), grad.clone()
        t1 = timeit(lambda: ops.sdf_mlp(plan, vols, pts, sdf_out=sdf, precision="f16x2"))
        t2 = timeit(lambda: ops.sdf_mlp(plan, vols, pts, want_grad=True, sdf_out=sdf, grad_out=grad, precision="f16x2"))
        print(f"L={len(dims)} f16x2 fwd : {t1:7.2f} ms  {n / t1 / 1e3:7.1f} Mpts/s  (equiv {n * f_fwd / t1 / 1e9:6.1f} TFLOP/s)")
        print(f"L={len(dims)} f16x2 grad: {t2:7.2f} ms  {n / t2 / 1e3:7.1f} Mpts/s  (equiv {n * 2 * f_fwd / t2 / 1e9:6.1f} TFLOP/s)"
              f"  max|dsdf| {float((sdf - s32).abs().max()):.2e} max|dgrad| {float((grad - g32).abs().max()):.2e} overflow {plan.overflowed()}")
    sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
    views = ops.SceneViews(sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev), [f.to(dev) for f in sc["features"]])
    bp = ops.BlendPlan(surf.color_network)
    rgb = torch.zeros(n, 3, device=dev)
    vis = torch.zeros(n, 4, device=dev, dtype=torch.uint8)
    t3 = timeit(lambda: ops.blend_views(bp, views, pts, rgb_out=rgb, vis_out=vis))
    fl = 2 * 4 * (4 * 16 + 16 * 23 + 69 * 64 + 64 * 32 + 32 * 32 + 32 * 33 + 32 * 32 + 32 + 37 * 16 + 16 * 8 + 8)
    print(f"blend views : {t3:7.2f} ms  {n / t3 / 1e3:7.1f} Mpts/s  {n * fl / t3 / 1e9:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
