#!/usr/bin/env python3
"""CPU probe (this container only): gradients of one fine-tune / train step computed by the reference's own model (imported through
tests/golden/make_golden.py's shims) against the CPU oracle with the truncated sampler, on identical inputs, samples and draws.
Answers "is a gradient discrepancy of the HIP path a property of the oracle's restatement or of the device code?".

    python scripts/probe/oracle_vs_reference_training.py [finetune|train] [L]
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import make_golden as MG  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "finetune"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dims = (64, 32, 16, 8, 4)[:nl] if nl == 5 else (16, 8, 4)
MG._install_shims()
from gens_amd import synthetic  # noqa: E402
from gens_amd.config import gens_model_conf  # noqa: E402
from gens_amd.models.modules.feature_network import _mnasnet_trunk  # noqa: E402
from oracle import gens_oracle as K  # noqa: E402
from oracle import render_oracle as R  # noqa: E402

sys.modules["torchvision.models"].mnasnet1_0 = lambda pretrained=True: types.SimpleNamespace(
    layers=nn.Sequential(*_mnasnet_trunk(), nn.Identity(), nn.Identity(), nn.Identity()))
from models.gens import GenS  # noqa: E402

seed = 280
torch.manual_seed(seed)
model = GenS(MG.Conf(dict(gens_model_conf(volume_dims=dims)))).train()
h, w, nv, n_rays = 64, 96, 4, 16
sc = synthetic.make_scene(nv=nv, h=h, w=w, n_levels=1, seed=seed + 1)
model.init_volumes({"imgs": sc["imgs"], "intrs": sc["intrs"], "c2ws": sc["c2ws"]})
view_ids = [2, 0, 3]
g = torch.Generator().manual_seed(seed + 2)
pix = torch.stack([torch.randint(8, w - 8, (n_rays,), generator=g), torch.randint(8, h - 8, (n_rays,), generator=g)], -1)
intrs, c2ws = sc["intrs"][view_ids], sc["c2ws"][view_ids]
rays_o, rays_d = synthetic.make_rays(intrs, c2ws, h, w, pixels=pix)
ipts = {"imgs": sc["imgs"][view_ids], "intrs": intrs, "c2ws": c2ws, "rays_o": rays_o, "rays_d": rays_d, "near": sc["near"], "far": sc["far"],
        "pseudo_pts": torch.rand(64, 3, generator=g) - 0.5, "view_ids": view_ids}

rec, draws = {}, []
surf = model.implicit_surface
orig_core = surf.render_core


def core(ro, rd, z, *a, **k):
    rec["z"] = z.detach().clone()
    return orig_core(ro, rd, z, *a, **k)


surf.render_core = core
orig_rand = torch.rand


def rand(*a, **k):
    r = orig_rand(*a, **k)
    draws.append(r.clone())
    return r


def loss_of(out, pseudo):
    hit = out["mid_inside_sphere"].reshape(1, -1, 1, 1)
    return (out["color_fine"].abs().sum() + 0.1 * out["gradient_error"] + 0.01 * out["smooth_error"] + 0.01 * out["tv_reg"]
            + torch.exp(-out["sparse_sdf"].abs() * 100).mean() + (((out["sampled_gray_val"] - out["ref_gray_val"]) ** 2) * hit).mean()
            + 0.1 * out["render_depth"].sum() + pseudo.abs().mean())


torch.rand = rand
torch.manual_seed(seed + 3)
try:
    out = model("finetune", ipts, cos_anneal_ratio=1.0, step=11)
finally:
    torch.rand = orig_rand
loss = loss_of(out, out["pseudo_sdf"])
loss.backward()
ref_grads = {k: p.grad.clone() for k, p in surf.named_parameters() if p.grad is not None}
ref_vgrads = [v.grad.clone() for v in model.volumes]

sd = {k: v.detach().clone().requires_grad_(True) for k, v in surf.state_dict().items()}
vols = [v.detach().clone().requires_grad_(True) for v in model.volumes]
masks = [m.detach() for m in model.mask_volmes]
feats = [f.detach()[view_ids] for f in model.features]
o = R.render(sd, rays_o, rays_d, sc["near"], sc["far"], vols, masks, ipts["imgs"], feats, feats, intrs, c2ws, 1.0, 11, draws[0], draws[1] * 2 - 1,
             truncated=True, z=rec["z"])
pp = ipts["pseudo_pts"]
okp = K.point_valid(masks, pp)
pseudo = torch.zeros(pp.shape[0], 1)
pseudo[okp] = R.sdf_mlp(sd, pp[okp], vols, K.lookup_volume_truncated)[:, :1]
lo = loss_of(o, pseudo)
lo.backward()
print("loss", float(loss), float(lo))
rows = []
for k, gref in ref_grads.items():
    a = sd[k].grad
    rows.append(((a - gref).abs().max().item() / max(gref.abs().max().item(), 1e-30), gref.abs().max().item(), k))
for i, (a, b) in enumerate(zip(vols, ref_vgrads)):
    rows.append(((a.grad - b).abs().max().item() / max(b.abs().max().item(), 1e-30), b.abs().max().item(), f"volume{i}"))
for e, m, k in sorted(rows, reverse=True)[:16]:
    print(f"{e:9.2e}  |max| {m:9.2e}  {k}")
