#!/usr/bin/env python3
"""One ray of the 1152 x 1600 fine-tune test (default: ray 90, an image-corner pixel): K18 (ops.blend_train) against the oracle's lookup_feature +
blend_mlp on that ray's section mid-points -- colours, visibility flags and the parameter gradients of sum(colour * W)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import ops, synthetic  # noqa: E402
from oracle import gens_oracle as K  # noqa: E402
from oracle import render_oracle as R  # noqa: E402
from tests.test_hip_shipped_shapes import _scene, _surface  # noqa: E402
from gens_amd.models.modules.implicit_surface import Scene  # noqa: E402

h, w = 1152, 1600
sc = _scene(3, h, w, seed=40)
surf = _surface(2).cuda().train()
g = torch.Generator().manual_seed(9)
pix = torch.stack([torch.randint(0, w, (512,), generator=g), torch.randint(0, h, (512,), generator=g)], -1)
ro, rd = synthetic.make_rays(sc["cpu"]["intrs"], sc["cpu"]["c2ws"], h, w, pixels=pix)
t_rand = torch.rand(512, 1, generator=g)
ray = int(sys.argv[1]) if len(sys.argv) > 1 else 90
scene = Scene(sc["vols"], sc["masks"], sc["imgs"], sc["features"], sc["features"], sc["intrs"], sc["c2ws"])
idx = torch.tensor([ray])
with torch.no_grad():
    z0 = sc["near"] + (sc["far"] - sc["near"]) * torch.linspace(0, 1, 64).cuda()[None]
    z0 = (z0.expand(1, 64) + (t_rand[idx].cuda() - 0.5) * 2.0 / 64).contiguous()
    z = surf._sample_rays(ro[idx].cuda().contiguous(), rd[idx].cuda().contiguous(), z0, scene)
    pts, valid = ops.ray_points(ro[idx].cuda().contiguous(), rd[idx].cuda().contiguous(), z, scene.masks, mid=True, sample_dist=2.0 / 64)
pts = pts.reshape(-1, 3)
keep = valid.reshape(-1).bool()
pts = pts[keep].contiguous()
print("ray", ray, "pixel", pix[ray].tolist(), "mid-points inside the masks:", int(keep.sum()))
W = torch.rand(pts.shape[0], 3, generator=g)
for p in surf.parameters():
    p.grad = None
rgb, vis = ops.blend_train(surf.color_network, scene.views, pts)
(rgb * W.cuda()).sum().backward()
cpu = sc["cpu"]
sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in surf.state_dict().items()}
fv, rdiff, mask = K.lookup_feature(pts.cpu(), cpu["imgs"], cpu["intrs"], cpu["c2ws"], cpu["features"])
col = R.blend_mlp(sd, fv, rdiff, mask)
(col * W).sum().backward()
print("visibility flags differ at", int((vis.cpu().bool() != mask).sum()), "of", mask.numel(), "; views visible per point (oracle):", torch.bincount(mask.sum(1), minlength=3).tolist())
print("colour max |diff|", float((rgb.detach().cpu() - col.detach()).abs().max()))
for k, p in surf.color_network.named_parameters():
    a, b = p.grad.detach().cpu(), sd["color_network." + k].grad
    print("%-22s dev max %.3e  oracle max %.3e  diff %.3e" % (k, float(a.abs().max()), float(b.abs().max()), float((a - b).abs().max())))
d = (rgb.detach().cpu() - col.detach()).abs().max(1).values
for k in range(3):
    sel = mask.sum(1) == k
    if sel.any():
        i = int(torch.argmax(d * sel))
        print("points with %d visible views: %d, colour max |diff| %.3e; worst: dev %s oracle %s" % (k, int(sel.sum()), float(d[sel].max()), rgb[i].tolist(), col[i].tolist()))
