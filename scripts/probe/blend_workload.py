#!/usr/bin/env python3
"""A few launches of the fused blending kernel (gens_blend_views4) on 4 M points, for counter passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import ops, synthetic  # noqa: E402
from gens_amd.config import gens_model_conf  # noqa: E402
from gens_amd.models.modules.implicit_surface import ImplicitSurface  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 1024 * 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
surf = ImplicitSurface(gens_model_conf(volume_dims=(256, 128, 64))["implicit_surface"]).to(dev).eval()
views = ops.SceneViews(sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev), [f.to(dev) for f in sc["features"]])
pts = torch.rand(n, 3, device=dev) * 1.6 - 0.8
plan = ops.BlendPlan(surf.color_network)
rgb = torch.zeros(n, 3, device=dev)
vis = torch.zeros(n, 4, dtype=torch.uint8, device=dev)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops.blend_views(plan, views, pts, rgb_out=rgb, vis_out=vis)
torch.cuda.synchronize()
s.record()
for _ in range(5):
    ops.blend_views(plan, views, pts, rgb_out=rgb, vis_out=vis)
e.record()
torch.cuda.synchronize()
print(f"blend {n} points: {s.elapsed_time(e) / 5:.3f} ms")
