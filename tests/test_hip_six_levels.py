"""More volume levels than the fused SDF kernels are built for (sdf_network.py:28-95 takes any `feat_channels`; the C ABI allows GENS_MAX_LEVELS = 8):
six levels run the PyTorch layers of SDFNetwork on the stand-alone look-up kernels K2 / K2'' -- inference and training -- and must agree with the
CPU oracle like every other configuration."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

DIMS = (16, 12, 8, 8, 4, 4)


def _setup():
    from gens_amd import synthetic
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.implicit_surface import ImplicitSurface
    sc = synthetic.make_scene(nv=4, h=48, w=64, n_levels=5, seed=11)
    torch.manual_seed(4)
    surf = ImplicitSurface(gens_model_conf(volume_dims=DIMS)["implicit_surface"])
    with torch.no_grad():                                         # (relative perturbation of the geometric initialisation: the volumes matter, a surface stays)
        for p in surf.sdf_network.parameters():
            p.mul_(1.0 + 0.05 * torch.randn_like(p))
    vols = synthetic.make_volumes(list(DIMS), seed=3)
    g = torch.Generator().manual_seed(6)
    masks = [(torch.rand(1, 1, d, d, d, generator=g) > 0.25).float() for d in DIMS]
    pix = torch.stack([torch.randint(8, 56, (24,), generator=g), torch.randint(8, 40, (24,), generator=g)], -1)
    ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 48, 64, pixels=pix)
    t_rand, pts_rand = torch.rand(24, 1, generator=g), torch.rand(1024, 3, generator=g) * 2 - 1
    return sc, surf, vols, masks, ro, rd, t_rand, pts_rand


def test_six_levels_take_the_layer_path_and_match_the_oracle_forward_and_backward():
    from gens_amd import ops
    from oracle import render_oracle as R
    from tests.test_hip_training import _loss
    sc, surf, vols, masks, ro, rd, t_rand, pts_rand = _setup()
    assert surf.sdf_network.init_feat_channels == 24 and not ops.SdfMlpPlan.supported(surf.sdf_network) or len(DIMS) > 5
    # oracle: the same weights, differentiable, second order truncated like the reference's Function pair
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in surf.state_dict().items()}
    vols_c = [v.clone().requires_grad_(True) for v in vols]
    ref = R.render(sd, ro, rd, sc["near"], sc["far"], vols_c, masks, sc["imgs"], sc["features"], sc["features"], sc["intrs"], sc["c2ws"], 0.5, None,
                   t_rand, pts_rand, truncated=True)
    _loss(ref).backward()
    surf = surf.cuda()
    c = lambda t: t.cuda()  # noqa: E731
    feats = [c(f) for f in sc["features"]]
    vols_d = [c(v).requires_grad_(True) for v in vols]
    masks_d = [c(m) for m in masks]
    launched = set()
    from gens_amd import lib as L
    L.profile_begin()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = surf.render(c(ro), c(rd), c(sc["near"]), c(sc["far"]), vols_d, masks_d, c(sc["imgs"]), feats, feats, c(sc["intrs"]), c(sc["c2ws"]), 0.5, None,
                          t_rand=t_rand, pts_random=c(pts_rand))
        _loss(out).backward()
    launched = set(L.profile_end())
    assert "gens_lookup_volume_bwd2" in launched and not any(k.startswith("gens_sdf_train") or k in ("gens_sdf_value", "gens_sdf_grad") for k in launched), launched
    # forward: the north-star bound on colour / depth, every per-ray output
    assert (out["color_fine"].cpu() - ref["color_fine"]).abs().mean() < 1e-4
    assert (out["render_depth"].cpu() - ref["render_depth"]).abs().mean() < 1e-4
    assert (out["sdf_depth"].cpu() - ref["sdf_depth"]).abs().mean() < 1e-4
    for k in ("weights", "gradients", "normal", "gradient_error", "smooth_error", "sparse_sdf"):
        a, b = out[k].detach().cpu(), ref[k].detach()
        assert float((a - b).abs().max()) <= 2e-4 * max(float(b.abs().max()), 1.0), k
    # backward: every parameter of the three networks and all six volumes
    top = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for name, p in surf.named_parameters():
        assert p.grad is not None, name
        r = sd[name].grad
        scale = max(float(r.abs().max()), 1e-4 * top)
        tol = 2e-3 if p.numel() > 4 else 0.15                        # (single scalars: sums of cancelling per-sample terms, as in test_hip_training.py)
        assert float((p.grad.cpu() - r).abs().max()) / scale < tol, name
    for i in range(len(DIMS)):
        r = vols_c[i].grad
        assert float((vols_d[i].grad.cpu() - r).abs().max()) <= 2e-3 * max(float(r.abs().max()), 1e-4 * top), i


def test_six_levels_inference_path_matches_the_oracle():
    from oracle import render_oracle as R
    sc, surf, vols, masks, ro, rd, t_rand, pts_rand = _setup()
    sd = {k: v.detach().clone() for k, v in surf.state_dict().items()}
    with torch.no_grad():
        ref = R.render(sd, ro, rd, sc["near"], sc["far"], vols, masks, sc["imgs"], sc["features"], sc["features"], sc["intrs"], sc["c2ws"], 1.0, None,
                       t_rand, pts_rand)
    surf = surf.cuda().eval()
    c = lambda t: t.cuda()  # noqa: E731
    feats = [c(f) for f in sc["features"]]
    with torch.no_grad():
        out = surf.render(c(ro), c(rd), c(sc["near"]), c(sc["far"]), [c(v) for v in vols], [c(m) for m in masks], c(sc["imgs"]), feats, feats, c(sc["intrs"]),
                          c(sc["c2ws"]), 1.0, None, lean=True, t_rand=t_rand)
    assert (out["color_fine"].cpu() - ref["color_fine"]).abs().mean() < 1e-4
    assert (out["render_depth"].cpu() - ref["render_depth"]).abs().mean() < 1e-4
    assert (out["sdf_depth"].cpu() - ref["sdf_depth"]).abs().mean() < 1e-4
