#!/usr/bin/env python3
"""Per-phase GPU time of one validate() chunk (development aid): where does a render step spend its time?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gens_amd import ops, synthetic  # noqa: E402
from gens_amd.config import gens_model_conf  # noqa: E402
from gens_amd.models.modules.implicit_surface import ImplicitSurface, Scene  # noqa: E402
from gens_amd.models.modules.projector import lookup_feature  # noqa: E402


class T:
    def __init__(self):
        self.t = {}

    def __call__(self, name):
        outer = self

        class C:
            def __enter__(self):
                torch.cuda.synchronize()
                self.t0 = time.perf_counter()

            def __exit__(self, *a):
                torch.cuda.synchronize()
                outer.t[name] = outer.t.get(name, 0) + (time.perf_counter() - self.t0) * 1e3
        return C()


def main():
    dev = torch.device("cuda:0")
    dims = [256, 128, 64]
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
    imgs, intrs, c2ws = sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev)
    feats = [f.to(dev) for f in sc["features"]]
    vols = [v.to(dev) for v in synthetic.make_volumes(dims, seed=100)]
    ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640)
    sel = slice(150 * 640, 150 * 640 + chunk)
    ro, rd = ro[sel].to(dev), rd[sel].to(dev)
    torch.manual_seed(0)
    surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"]).to(dev).eval()
    with torch.no_grad():
        _, masks = ops.volume_build(feats[:3], intrs, c2ws, dims)
    scene = Scene(vols, masks, imgs, feats, feats, intrs, c2ws)
    near, far = sc["near"].to(dev), sc["far"].to(dev)
    for it in range(2):
        tm = T()
        with torch.no_grad():
            b = ro.shape[0]
            z = (near + (far - near) * torch.linspace(0, 1, 64, device=dev)[None]).expand(b, 64).contiguous()
            with tm("sample_rays (4 rounds, fwd MLP on 112 pts/ray)"):
                z = surf._sample_rays(ro, rd, z, scene)
            vols_p = scene.volumes_nograd()
            with tm("ray_points+select"):
                pts, valid = ops.ray_points(ro, rd, z, scene.masks, mid=True, sample_dist=1 / 32)
                idx = surf._select(valid)
                pts_v = pts[idx]
            with tm("sdf fwd + d/dx"):
                plan = surf._fused_plan(vols_p)
                if plan is not None:
                    s_full = torch.full((b * 128, 1), 100.0, device=dev)
                    g_full = torch.zeros(b * 128, 3, device=dev)
                    ops.sdf_mlp(plan, vols_p, pts, index=idx, want_grad=True, sdf_out=s_full, grad_out=g_full)
                    s, g = s_full[idx], g_full[idx]
                else:
                    with torch.enable_grad():
                        x = pts_v.clone().requires_grad_(True)
                        s = surf.sdf_network.sdf(x, vols_p)
                        g = torch.autograd.grad(s, x, torch.ones_like(s))[0]
            with tm("lookup_feature (K4)"):
                fv, rdiff, vis = lookup_feature(pts_v, imgs, intrs, c2ws, feats, views=scene.views)
            with tm("blend MLP"):
                col = surf.color_network(fv, rdiff, vis)
            with tm("scatter + composite"):
                n = 128
                sdf = torch.full((b * n, 1), 100.0, device=dev).index_put((idx,), s.detach())
                grad = torch.zeros(b * n, 3, device=dev).index_put((idx,), g)
                colf = torch.zeros(b * n, 3, device=dev).index_put((idx,), col)
                sv = torch.zeros(b * n, vis.shape[1], dtype=torch.bool, device=dev).index_put((idx,), vis)
                out = ops.composite(ro, rd, z, 1 / 32, sdf, grad, None, colf, valid, sv, torch.tensor([20.0], device=dev), 1.0, c2ws[0])
        if it == 1:
            tot = sum(tm.t.values())
            print(f"chunk={chunk} rays, valid fraction {idx.numel() / (b * 128):.3f}, total {tot:.1f} ms "
                  f"-> {chunk * 128 / tot * 1e3 / 1e6:.1f} M ray-samples/s")
            for k, v in tm.t.items():
                print(f"  {v:8.1f} ms  {100 * v / tot:5.1f}%  {k}")


if __name__ == "__main__":
    main()
