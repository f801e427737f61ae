#!/usr/bin/env python3
"""Which rays carry the colour-network gradient difference of tests/test_hip_shipped_shapes.py::test_finetune_step_three_views_1152x1600 at nb = 128?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import synthetic  # noqa: E402
from oracle import render_oracle as R  # noqa: E402
from tests.test_hip_shipped_shapes import _finetune_loss, _scene, _surface  # noqa: E402

h, w = 1152, 1600
sc = _scene(3, h, w, seed=40)
surf = _surface(2).cuda().train()
g = torch.Generator().manual_seed(9)
pix = torch.stack([torch.randint(0, w, (512,), generator=g), torch.randint(0, h, (512,), generator=g)], -1)
ro, rd = synthetic.make_rays(sc["cpu"]["intrs"], sc["cpu"]["c2ws"], h, w, pixels=pix)
t_rand = torch.rand(512, 1, generator=g)
pts_rand = torch.rand(1024, 3, generator=g) * 2 - 1
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
from gens_amd.models.modules.implicit_surface import Scene  # noqa: E402
scene = Scene(sc["vols"], sc["masks"], sc["imgs"], sc["features"], sc["features"], sc["intrs"], sc["c2ws"])
cpu = sc["cpu"]
masks_c = [m.cpu() for m in sc["masks"]]


def run(rays):
    n = len(rays)
    idx = torch.tensor(rays)
    with torch.no_grad():
        z0 = sc["near"] + (sc["far"] - sc["near"]) * torch.linspace(0, 1, 64).cuda()[None]
        z0 = (z0.expand(n, 64) + (t_rand[idx].cuda() - 0.5) * 2.0 / 64).contiguous()
        z = surf._sample_rays(ro[idx].cuda().contiguous(), rd[idx].cuda().contiguous(), z0, scene)
    for p in surf.parameters():
        p.grad = None
    vols_d = [v.detach().clone().requires_grad_(True) for v in sc["vols"]]
    out = surf.render_core(ro[idx].cuda().contiguous(), rd[idx].cuda().contiguous(), z, 2.0 / 64, vols_d, sc["masks"], sc["features"], sc["features"],
                           sc["imgs"], sc["intrs"], sc["c2ws"], 1.0, 11.0, pts_random=pts_rand.cuda())
    _finetune_loss(out).backward()
    sd_ = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in surf.state_dict().items()}
    vols_ = [v.clone().requires_grad_(True) for v in sc["vols_cpu"]]
    r = R.render(sd_, ro[idx], rd[idx], cpu["near"], cpu["far"], vols_, masks_c, cpu["imgs"], cpu["features"], cpu["features"], cpu["intrs"], cpu["c2ws"],
                 1.0, 11.0, t_rand[idx], pts_rand, truncated=True, z=z.cpu())
    _finetune_loss(r).backward()
    return out, r, {k: p.grad.detach().cpu() for k, p in surf.named_parameters()}, {k: v.grad for k, v in sd_.items()}


out, ref, gd, go = run(list(range(nb)))
dcol = (out["color_fine"].detach().cpu() - ref["color_fine"].detach()).abs().max(1).values
cd, co = out["color_fine"].detach().cpu(), ref["color_fine"].detach()
flip = (torch.sign(cd) != torch.sign(co))
print("sign flips of a colour channel (the test loss is |colour|.sum()):", int(flip.sum()), [(int(i), int(j), float(cd[i, j]), float(co[i, j])) for i, j in flip.nonzero()[:6]])
order = torch.argsort(dcol, descending=True)
print("worst rays by colour:", [(int(i), float(dcol[i])) for i in order[:8]])
for k in ("color_network.base_fc.0.bias", "color_network.ray_dir_fc.2.bias", "color_network.rgb_fc.4.weight"):
    print(k, float((gd[k] - go[k]).abs().max() / go[k].abs().max()))
keep = [int(i) for i in range(nb) if dcol[i] < 1e-5]
print("rays kept:", len(keep), "dropped:", [int(i) for i in range(nb) if dcol[i] >= 1e-5])
out2, ref2, gd2, go2 = run(keep)
for k in ("color_network.base_fc.0.bias", "color_network.ray_dir_fc.2.bias", "color_network.rgb_fc.4.weight"):
    print("without them:", k, float((gd2[k] - go2[k]).abs().max() / go2[k].abs().max()))

key = "color_network.base_fc.0.bias"
bad_chunk = None
for s0 in range(48, nb, 16):
    _, _, a, b = run(list(range(s0, min(nb, s0 + 16))))
    e = float((a[key] - b[key]).abs().max() / b[key].abs().max())
    print("rays", s0, s0 + 16, e, flush=True)
    if e > 1e-3 and bad_chunk is None:
        bad_chunk = s0
if bad_chunk is not None:
    for r in range(bad_chunk, bad_chunk + 16):
        o_, r_, a, b = run([r])
        e = float((a[key] - b[key]).abs().max() / b[key].abs().max())
        print("ray", r, "pixel", pix[r].tolist(), e, "colour", o_["color_fine"].detach().cpu().tolist(), "wsum", float(o_["weight_sum"].detach().cpu()), flush=True)
