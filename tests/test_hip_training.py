"""GPU tests of the training path (BASELINE config 3 / 5): backward through the HIP look-up (first + second order),
compositing, feature-warp, patch-warp and TV kernels; GenS.forward in train and fine-tune mode with stand-in CNNs."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _loss(out):
    hit = out["mid_inside_sphere"].reshape(1, -1, 1, 1)
    mfc = (((out["sampled_gray_val"] - out["ref_gray_val"]) ** 2) * hit).mean()
    return (out["color_fine"].abs().sum() + 0.1 * out["gradient_error"] + 0.01 * out["smooth_error"] + 0.01 * out["tv_reg"]
            + torch.exp(-out["sparse_sdf"].abs() * 100).mean() + mfc + 0.1 * out["render_depth"].sum() + 0.05 * out["normal"].sum())


def test_render_gradients_match_oracle(golden):
    """d loss / d {SDF-MLP, colour-MLP, variance, volumes, feature maps} : HIP path vs the CPU oracle whose sampler is
    truncated at second order exactly like the reference's Function pair (cuda_gridsample.py:110-123)."""
    from oracle import render_oracle as R
    from tests.test_hip_render import build_surface, scene_inputs
    g = golden("g9a_render")
    feats_c = [g[f"feat{i}"].clone().requires_grad_(True) for i in range(5)]
    vols_c = [g[f"vol{i}"].clone().requires_grad_(True) for i in range(3)]
    masks_c = [g[f"mask{i}"] for i in range(3)]
    sd = {k[3:]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith("sd.")}
    pts_rand = g["draw_ptsrand"] * 2 - 1
    ref = R.render(sd, g["rays_o"], g["rays_d"], g["near"], g["far"], vols_c, masks_c, g["imgs"], feats_c, feats_c, g["intrs"], g["c2ws"],
                   0.5, None, g["draw_trand"], pts_rand, truncated=True, z=g["z_final"])
    _loss(ref).backward()

    surf = build_surface(g)
    c = lambda t: t.cuda()  # noqa: E731
    feats = [c(g[f"feat{i}"]).requires_grad_(True) for i in range(5)]
    vols = [c(g[f"vol{i}"]).requires_grad_(True) for i in range(3)]
    masks = [c(m) for m in masks_c]
    out = surf.render_core(c(g["rays_o"]), c(g["rays_d"]), c(g["z_final"]), 2.0 / 64, vols, masks, feats, feats, c(g["imgs"]), c(g["intrs"]),
                           c(g["c2ws"]), 0.5, None, pts_random=c(pts_rand))
    _loss(out).backward()

    top = max(v.grad.abs().max().item() for v in sd.values() if v.grad is not None)

    def check(name, a, b, tol=2e-3):
        a, b = a.detach().cpu(), b.detach().cpu()
        # gradients that are zero by symmetry (e.g. the bias in front of a softmax) are pure round-off on both sides
        scale = max(b.abs().max().item(), 1e-4 * top)
        err = (a - b).abs().max().item() / scale
        assert err < tol, f"{name}: relative-to-max error {err:.2e}"
    for name, p in surf.named_parameters():
        assert p.grad is not None, name
        # single scalars (anti-alias temperature, variance) are sums of cancelling per-sample terms: compare loosely
        check(name, p.grad, sd[name].grad, tol=2e-3 if p.numel() > 4 else 0.15)
    for i in range(3):
        check(f"volume{i}", vols[i].grad, vols_c[i].grad)
    for i in range(5):
        check(f"feature{i}", feats[i].grad, feats_c[i].grad)


class TinyFeatureNet(nn.Module):
    """Stand-in for the MnasNet encoder/decoder (out of scope): 5-level pyramid, 4 channels each."""

    def __init__(self, confs):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(3, 4, 3, padding=1) for _ in range(5)])

    def forward(self, imgs):
        outs, x = [], imgs
        for i, conv in enumerate(self.convs):
            outs.append(conv(x))
            x = nn.functional.avg_pool2d(x, 2)
        return outs


class TinyRegNet(nn.Module):
    """Stand-in for the 3-D U-Net (out of scope): one 1x1x1 conv per level, 8 -> 4 channels."""

    def __init__(self, confs):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv3d(8, 4, 1) for _ in confs.get_list("d_out")])

    def forward(self, volumes):
        return [conv(v) for conv, v in zip(self.convs, volumes)]


def _gens(dims=(16, 8, 4)):
    from gens_amd.config import gens_model_conf
    from gens_amd.models import gens
    gens.register_backbones(TinyFeatureNet, TinyRegNet)
    torch.manual_seed(0)
    return gens.GenS(gens_model_conf(volume_dims=dims)).cuda()


def _inputs(n_rays=64, nv=4):
    from gens_amd import synthetic
    sc = synthetic.make_scene(nv=nv, h=48, w=64, n_levels=1, seed=5)
    g = torch.Generator().manual_seed(2)
    pix = torch.stack([torch.randint(4, 60, (n_rays,), generator=g), torch.randint(4, 44, (n_rays,), generator=g)], -1)
    ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 48, 64, pixels=pix)
    ipts = {"imgs": sc["imgs"], "intrs": sc["intrs"], "c2ws": sc["c2ws"], "rays_o": ro, "rays_d": rd, "near": sc["near"], "far": sc["far"],
            "pseudo_pts": torch.rand(256, 3, generator=g) - 0.5}
    return {k: v.cuda() for k, v in ipts.items()}


def test_gens_train_step_backward_reaches_every_trainable_parameter():
    model = _gens().train()
    ipts = _inputs()
    out = model("train", ipts, cos_anneal_ratio=0.3, step=1.0)
    assert set(out) >= {"color_fine", "gradient_error", "sparse_sdf", "smooth_error", "tv_reg", "ref_gray_val", "sampled_gray_val",
                        "mid_inside_sphere", "pseudo_sdf", "render_depth", "valid_mask", "weights", "gradients", "normal", "s_val",
                        "weight_sum", "weight_max", "inside_sphere", "sdf_depth"}
    loss = _loss(out) + out["pseudo_sdf"].abs().mean()
    loss.backward()
    for name, p in model.named_parameters():
        if name.startswith("match_feature_network"):
            assert p.grad is None                      # frozen copy (gens.py:22-24)
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    assert model.feature_network.convs[0].weight.grad.abs().sum() > 0      # through K4 (and K1 -> reg net -> K2)
    assert model.feature_network.convs[2].weight.grad.abs().sum() > 0      # level 2 feeds K1 and K4
    assert model.reg_network.convs[0].weight.grad.abs().sum() > 0          # through K2 / K2'' / K10
    opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3}))
    opt.step()


def test_gens_stand_alone_train_step_with_its_own_backbones():
    """GenS with this package's FeatureNetwork (MnasNet trunk) and RegNetwork (3-D U-Net), nothing registered and no reference tree:
    one train step at 64x96, gradients through K4 / K1 into the 2-D CNN and through K2 / K2'' / K10 into the 3-D one."""
    from gens_amd import synthetic
    from gens_amd.config import gens_model_conf
    from gens_amd.models import gens
    saved = dict(gens._BACKBONES)
    gens._BACKBONES.clear()
    try:
        torch.manual_seed(0)
        model = gens.GenS(gens_model_conf(volume_dims=(16, 8, 4))).cuda().train()
    finally:
        gens._BACKBONES.update(saved)
    sc = synthetic.make_scene(nv=4, h=64, w=96, n_levels=1, seed=5)
    g = torch.Generator().manual_seed(2)
    pix = torch.stack([torch.randint(4, 92, (64,), generator=g), torch.randint(4, 60, (64,), generator=g)], -1)
    ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 64, 96, pixels=pix)
    ipts = {k: v.cuda() for k, v in {"imgs": sc["imgs"], "intrs": sc["intrs"], "c2ws": sc["c2ws"], "rays_o": ro, "rays_d": rd, "near": sc["near"],
                                     "far": sc["far"], "pseudo_pts": torch.rand(256, 3, generator=g) - 0.5}.items()}
    out = model("train", ipts, cos_anneal_ratio=0.3, step=5)               # step % 5 == 0: the matching network takes the weights (gens.py:133-138)
    (_loss(out) + out["pseudo_sdf"].abs().mean()).backward()
    for name, p in model.named_parameters():
        if name.startswith("match_feature_network"):
            assert p.grad is None
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    assert model.feature_network.layer1[0].weight.grad.abs().sum() > 0 and model.feature_network.out_layer5.weight.grad.abs().sum() > 0
    assert model.reg_network.conv0.conv.weight.grad.abs().sum() > 0 and model.reg_network.out_layers[2].weight.grad.abs().sum() > 0
    for a, b in zip(model.feature_network.parameters(), model.match_feature_network.parameters()):
        assert torch.equal(a, b)                       # (the BatchNorm running statistics have moved on with the matching pass)
    torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3})).step()


def test_gens_finetune_volumes_are_parameters_and_checkpoint_roundtrip(tmp_path):
    model = _gens()
    ipts = _inputs(nv=3)
    model.init_volumes({k: ipts[k] for k in ("imgs", "intrs", "c2ws")})
    assert model.has_vol and len(model.volumes) == 3 and not model.mask_volmes[0].requires_grad
    groups = model.get_optim_params({"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]})
    assert len(groups) == 4
    ipts["view_ids"] = [0, 1, 2]
    out = model("finetune", ipts, 1.0, None)
    _loss(out).backward()
    for v in model.volumes:
        assert v.grad is not None and v.grad.abs().sum() > 0
    path = tmp_path / "vol.ckpt"
    torch.save({"model": model.get_params_vol()}, path)
    from gens_amd.config import gens_model_conf
    from gens_amd.models import gens
    fresh = gens.GenS(gens_model_conf(volume_dims=(16, 8, 4), has_vol=True)).cuda()
    fresh.load_params_vol(str(path), "cuda")
    with torch.no_grad():
        torch.manual_seed(4)                           # same ray jitter in both renders
        a = fresh("finetune", ipts, 1.0, None)["sparse_sdf"][1024:]
        torch.manual_seed(4)
        b = model("finetune", ipts, 1.0, None)["sparse_sdf"][1024:]
    assert torch.allclose(a, b, atol=1e-5)


def test_gens_val_mode_returns_image_buffers():
    model = _gens().eval()
    ipts = _inputs(n_rays=48)
    ipts.update(bound_min=torch.tensor([-1.0, -1, -1]).cuda(), bound_max=torch.tensor([1.0, 1, 1]).cuda(), hw=torch.tensor([6, 8]).int())
    ipts.pop("pseudo_pts")
    surf = model.implicit_surface
    with torch.no_grad():
        feats = model.feature_network(ipts["imgs"])
        vols, masks = model.volume.agg_mean_var(feats, ipts["intrs"], ipts["c2ws"])
        vols = model.reg_network(vols)
        out = surf.validate(ipts["rays_o"], ipts["rays_d"], ipts["near"], ipts["far"], vols, masks, ipts["imgs"], feats, feats, ipts["intrs"],
                            ipts["c2ws"], ipts["bound_min"], ipts["bound_max"], ipts["hw"], extract_geometry=False)
    assert out["img_fine"].shape == (6, 8, 3) and out["normal_img"].shape == (6, 8, 3)
    assert out["sdf_depth"].shape == (6, 8) and out["render_depth"].shape == (6, 8) and tuple(out["color_fine"].shape) == (48, 3)


def test_gens_forward_val_returns_mesh_and_images():
    """GenS.forward('val') end to end (gens.py:86-122 -> implicit_surface.py:429-470): the 7 output keys of the reference,
    with the 512^3 mesh extraction running on the device (no PyMCubes)."""
    model = _gens().eval()
    ipts = _inputs(n_rays=48)
    ipts.update(bound_min=torch.tensor([-1.0, -1, -1]).cuda(), bound_max=torch.tensor([1.0, 1, 1]).cuda(), hw=torch.tensor([6, 8]).int())
    ipts.pop("pseudo_pts")
    with torch.no_grad():
        out = model("val", ipts, 1.0, None)
    assert set(out) >= {"vertices", "triangles", "color_fine", "img_fine", "normal_img", "sdf_depth", "render_depth"}
    v, t = out["vertices"], out["triangles"]
    assert v.ndim == 2 and v.shape[1] == 3 and t.ndim == 2 and t.shape[1] == 3 and v.dtype.kind == "f"
    if len(t):
        assert t.min() >= 0 and t.max() < len(v) and np.abs(v).max() <= 1.0 + 1e-6


def test_lncc_on_render_outputs_backpropagates_like_the_oracle():
    """loss.py:36-38 on the patch tensors render_core returns: K13 forward value and the gradient reaching the patch tensors."""
    from gens_amd.losses import compute_LNCC
    from oracle import gens_oracle as K
    model = _gens().train()
    ipts = _inputs(n_rays=32)
    out = model("train", ipts, 1.0, None)
    ref, src = out["ref_gray_val"], out["sampled_gray_val"]
    assert ref.shape[0] == 1 and ref.shape[2:] == (121, 12) and src.shape[1:] == ref.shape[1:]
    ncc = compute_LNCC(ref, src)
    mask = out["valid_mask"] * out["mid_inside_sphere"]
    loss = 0.5 * ((ncc * mask).sum(0) / (mask.sum(0) + 1e-8)).squeeze(-1)
    wrt = [t for t in (ref, src) if t.requires_grad]
    assert src.requires_grad                                   # the patch samples depend on the SDF network through the surface point
    got = dict(zip([id(t) for t in wrt], torch.autograd.grad(loss, wrt, retain_graph=True)))
    r_o, s_o = ref.detach().cpu().requires_grad_(True), src.detach().cpu().requires_grad_(True)
    want = K.lncc(r_o, s_o)
    assert torch.allclose(ncc.detach().cpu(), want.detach(), atol=1e-5)
    loss_o = 0.5 * ((want * mask.detach().cpu()).sum(0) / (mask.detach().cpu().sum(0) + 1e-8)).squeeze(-1)
    w_ref, w_src = torch.autograd.grad(loss_o, [r_o, s_o])
    scale = float(w_src.abs().max()) + 1e-12
    assert float((got[id(src)].cpu() - w_src).abs().max()) <= 1e-3 * scale + 1e-9
    if ref.requires_grad:
        assert float((got[id(ref)].cpu() - w_ref).abs().max()) <= 1e-3 * scale + 1e-9
    loss.backward()                                            # reaches the colour-independent path: features / intrinsics side is detached, MLPs get grads
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_dataset_item_drives_train_and_val_end_to_end(tmp_path):
    """Caller side of the path: a DTUDataset item goes through GenS.forward unchanged, as in runner.py:150-160."""
    from gens_amd.config import Conf
    from gens_amd.datasets import DTUDataset
    from tests import dtu_fixture
    root = dtu_fixture.make_dtu_tree(str(tmp_path / "dtu"))
    model = _gens().train()
    to_dev = lambda item: {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in item.items()}  # noqa: E731
    conf = dtu_fixture.conf_values(root, "train")
    conf["img_hw"] = [64, 80]                                  # five pyramid levels need sizes divisible by 16
    torch.manual_seed(1)
    item = to_dev(DTUDataset(Conf(conf), "train")[0])
    out = model("train", item, 0.5, 10.0)
    assert out["color_fine"].shape == (64, 3) and out["pseudo_sdf"].shape == (2048, 1)
    loss = _loss(out) + out["pseudo_sdf"].abs().mean()
    loss.backward()
    assert torch.isfinite(loss)
    conf = dtu_fixture.conf_values(root, "val")
    conf["img_hw"] = [64, 80]
    conf["val_res_level"] = 4
    item = to_dev(DTUDataset(Conf(conf), "val")[0])
    with torch.no_grad():
        val = model.eval()("val", item, 1.0, None)
    assert val["img_fine"].shape == (16, 20, 3) and val["render_depth"].shape == (16, 20) and val["vertices"].shape[1] == 3


def test_finetune_dataset_drives_the_per_scene_path(tmp_path):
    """BASELINE config 5 plumbing as runner.py does it (:88-96, :294-300, :346): init_volumes from get_all_images, a fine-tune step on
    get_random_rays, validation rays from get_rays_at."""
    from gens_amd.config import Conf
    from gens_amd.datasets import DTUDatasetFinetune
    from tests import dtu_fixture
    root = dtu_fixture.make_dtu_tree(str(tmp_path / "dtu"))
    conf = dtu_fixture.finetune_conf_values(root)
    conf["img_hw"] = [64, 80]
    ds = DTUDatasetFinetune(Conf(conf), "finetune")
    model = _gens()
    dev = lambda item: {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in item.items()}  # noqa: E731
    model.init_volumes(dev(ds.get_all_images()))
    assert model.has_vol and len(model.volumes) == 3
    out = model("finetune", dev(ds.get_random_rays(torch.tensor(1))), 1.0, None)
    loss = _loss(out) + out["pseudo_sdf"].abs().mean()
    loss.backward()
    assert torch.isfinite(loss) and all(v.grad is not None for v in model.volumes)
    item = dev(ds.get_rays_at(0))
    with torch.no_grad():
        val = model.eval()("val", item, 1.0, None)
    assert val["img_fine"].shape == (32, 40, 3)


def test_bmvs_datasets_drive_val_finetune_and_the_writers(tmp_path):
    """BlendedMVS front-ends (BASELINE config 5's evaluation set) through the same plumbing: a BMVSDataset val item through
    GenS.forward("val"), the per-scene fine-tune path on BMVSDatasetFinetune (no pseudo points there), and the validation outputs
    stored the way runner.py:229-246 stores them."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import bmvs_fixture
    from gens_amd import io
    from gens_amd.config import Conf
    from gens_amd.datasets import BMVSDataset, BMVSDatasetFinetune
    root = bmvs_fixture.make_bmvs_tree(str(tmp_path / "bmvs"))
    dev = lambda item: {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in item.items()}  # noqa: E731
    conf = bmvs_fixture.conf_values(root, "val")
    conf["img_hw"] = [64, 80]
    conf["val_res_level"] = 4
    item = dev(BMVSDataset(Conf(conf), "val")[0])
    model = _gens().eval()
    with torch.no_grad():
        val = model("val", item, 1.0, None)
    assert val["img_fine"].shape == (16, 20, 3) and val["sdf_depth"].shape == (16, 20) and val["triangles"].shape[1] == 3
    paths = io.save_validation_outputs(str(tmp_path / "exp"), val, item, "epoch0", clean=True)
    v, t = io.read_ply(paths["mesh"])
    assert len(t) <= len(val["triangles"]) and v.shape[0] == val["vertices"].shape[0] and os.path.getsize(paths["normal"]) > 0
    ft_conf = bmvs_fixture.finetune_conf_values(root)
    ft_conf["img_hw"] = [64, 80]
    ds = BMVSDatasetFinetune(Conf(ft_conf), "finetune")
    model = _gens()
    model.init_volumes(dev(ds.get_all_images()))
    out = model("finetune", dev(ds.get_random_rays(torch.tensor(2))), 1.0, None)
    assert "pseudo_sdf" not in out                                  # BlendedMVS has no pseudo-depth points (bmvs_finetune.py:233-266)
    loss = _loss(out)
    loss.backward()
    assert torch.isfinite(loss) and all(vol.grad is not None for vol in model.volumes)


# Gradient tolerances, as a fraction of each gradient tensor's largest magnitude.  Measured against the reference's own backward
# (goldens g17 / g17b / g18 / g18b, whose sampler drops third order like the reference's CUDA Function pair): implicit-surface
# parameters and fine-tune volumes 1e-5 ... 8e-4 (the reference's own 1-thread / 8-thread runs differ by 6e-6; the 8e-4 is lin1..6.weight_v
# of the five-level fine-tune golden g18b, where the volumes come out of `init_volumes` -- the device CNNs, within 2e-4 of ATen's -- and a
# 1e-4 perturbation of the volumes alone moves exactly those gradients by 1.2e-3: scripts/probe/oracle_vs_reference_training.py; with the
# reference's own volumes the CPU oracle reproduces them to 6e-6, and the K17 kernels the oracle to 5e-5), the two CNNs up to
# 1.6e-2 (first MnasNet convolution: batch-norm statistics over three 64 x 96 views amplify float32 round-off of MIOpen against ATen).
# Round 3: per parameter group, three times what is measured (scripts: pytest -s prints the tables; profiles/r03_grad_tables.txt):
#   2-D CNN / 3-D U-Net 1.6e-2 / 1.0e-3 -> 3e-2 (kept), colour network 1.7e-3 -> 5e-3, SDF network biases / weight_g 2.7e-4 -> 1e-3,
#   SDF weight_v 8e-5 -> 3e-4 -- except the five-level fine-tune golden g18b with its free-running samples: 3.0e-3 -> 1e-2 --,
#   fine-tune volumes 6e-5 -> 2e-4, the variance 6e-6 -> 1e-4.
GRAD_RTOL_CNN = 3e-2
GRAD_ZERO = 1e-5     # a tensor whose largest golden entry is below GRAD_ZERO x the largest gradient in the table is analytically zero
                     # (the bias in front of a softmax, a head whose output is unused): pure round-off on both sides, compared absolutely


def _grad_tol(name, loose_weight_v=False):
    if "_network." in name and "implicit_surface" not in name:
        return GRAD_RTOL_CNN
    if "color_network" in name:
        return 5e-3
    if "variance" in name:
        return 1e-4
    if name.startswith("volume"):
        return 2e-4
    if "weight_v" in name:
        return 1e-2 if loose_weight_v else 3e-4
    return 1e-3


def _grad_table(rows, loose_weight_v=False):
    """rows: (name, error relative to the tensor's largest magnitude, largest magnitude) -> the rows that matter, worst first; printed in
    full (pytest -s / on failure).  Analytically-zero tensors are re-scaled to the table's largest gradient."""
    top = max(m for _, _, m in rows)
    rows = [(k, e * m / (GRAD_ZERO * top) if m < GRAD_ZERO * top else e, m) for k, e, m in rows]
    rows = sorted(rows, key=lambda r: -r[1] / _grad_tol(r[0], loose_weight_v))
    print("\n".join(f"  {e:9.2e}  |max| {m:9.2e}  {k}" for k, e, m in rows))
    return rows


def _check_grad_table(rows, loose_weight_v=False):
    rows = _grad_table(rows, loose_weight_v)
    bad = [r for r in rows if r[1] >= _grad_tol(r[0], loose_weight_v)]
    assert not bad, bad[:6]


@pytest.mark.parametrize("tag,dims,seed", [("g17_gens_forward", (16, 8, 4), 170), ("g17b_gens_forward_l5", (64, 32, 16, 8, 4), 270)])
def test_gens_forward_matches_the_reference_model_end_to_end(tag, dims, seed):
    """The whole model against the reference's own `GenS.forward("train", ...)` run on the CPU (golden g17: its FeatureNetwork, Volume,
    RegNetwork and ImplicitSurface classes, make_golden.py g17): same seeded backbone weights, same implicit-surface weights, same
    host RNG stream -> the 19 outputs, the loss and parameter gradients in every part of the model.  Floating point through two CNNs
    (MIOpen / K15 / K16 against ATen's CPU convolutions), the volume build, ~250 MLP evaluations per ray and their derivatives:
    outputs within 2e-4 of each tensor's largest magnitude, gradients within the per-group bounds of _grad_tol (three times what is measured)."""
    import numpy as np
    from gens_amd.config import gens_model_conf
    from gens_amd.models import gens
    from .conftest import GOLDEN
    raw = np.load(os.path.join(GOLDEN, tag + ".npz"))
    g = {k: raw[k] for k in raw.files}
    saved = dict(gens._BACKBONES)
    gens._BACKBONES.clear()
    try:
        torch.manual_seed(seed)
        model = gens.GenS(gens_model_conf(volume_dims=dims)).train()
    finally:
        gens._BACKBONES.update(saved)
    sd = model.state_dict()
    names = [k for k in sd if not k.startswith("implicit_surface.")]
    assert names == list(g["backbone.keys"])
    np.testing.assert_allclose([float(sd[k].double().sum()) for k in names], g["backbone.sums"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose([float(sd[k].double().abs().sum()) for k in names], g["backbone.abs_sums"], rtol=1e-9, atol=1e-9)
    model.implicit_surface.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    model = model.cuda()
    ipts = {k[3:]: torch.from_numpy(v).cuda() for k, v in g.items() if k.startswith("in.")}
    torch.manual_seed(seed + 3)
    out = model("train", ipts, cos_anneal_ratio=0.7, step=3)
    hit = out["mid_inside_sphere"].reshape(1, -1, 1, 1)
    loss = (out["color_fine"].abs().sum() + 0.1 * out["gradient_error"] + 0.01 * out["smooth_error"] + 0.01 * out["tv_reg"]
            + torch.exp(-out["sparse_sdf"].abs() * 100).mean() + (((out["sampled_gray_val"] - out["ref_gray_val"]) ** 2) * hit).mean()
            + 0.1 * out["render_depth"].sum() + out["pseudo_sdf"].abs().mean())
    loss.backward()
    worst = {}
    for k, v in g.items():
        if not k.startswith("out."):
            continue
        a, b = out[k[4:]].detach().cpu().double().reshape(-1), torch.from_numpy(v).double().reshape(-1)
        assert a.shape == b.shape, k
        worst[k] = ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()
    bad = {k: e for k, e in worst.items() if e > 2e-4}
    assert not bad, bad
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    params = dict(model.named_parameters())
    rows = []
    for k, v in g.items():
        if k.startswith("grad."):
            a, b = params[k[5:]].grad.cpu().double(), torch.from_numpy(v).double()
            # (the bias of the finest output head has an analytically zero gradient here: 1e-11 of round-off on both sides)
            rows.append((k, ((a - b).abs().max() / b.abs().max().clamp_min(1e-8)).item(), b.abs().max().item()))
    _check_grad_table(rows)


@pytest.mark.parametrize("tag,dims,seed", [("g18_gens_finetune", (16, 8, 4), 180), ("g18b_gens_finetune_l5", (64, 32, 16, 8, 4), 280)])
def test_gens_finetune_path_matches_the_reference_model(tag, dims, seed):
    """`init_volumes` + `forward("finetune", ...)` against the reference's own run (golden g18, models/gens.py:63-85,141-155): the CNN
    outputs frozen into parameters (volumes, masks, feature maps), a step on a re-ordered subset of the views, gradients of the volume
    parameters."""
    import numpy as np
    from gens_amd.config import gens_model_conf
    from gens_amd.models import gens
    from .conftest import GOLDEN
    raw = np.load(os.path.join(GOLDEN, tag + ".npz"))
    g = {k: raw[k] for k in raw.files}
    nl = len(dims)
    saved = dict(gens._BACKBONES)
    gens._BACKBONES.clear()
    try:
        torch.manual_seed(seed)
        model = gens.GenS(gens_model_conf(volume_dims=dims)).train()
    finally:
        gens._BACKBONES.update(saved)
    model.implicit_surface.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    model = model.cuda()
    t = lambda k: torch.from_numpy(g[k]).cuda()  # noqa: E731

    def rel(a, b):
        a, b = a.detach().cpu().double().reshape(-1), torch.from_numpy(b).double().reshape(-1)
        return ((a - b).abs().max() / b.abs().max().clamp_min(1e-8)).item()

    model.init_volumes({"imgs": t("all.imgs"), "intrs": t("all.intrs"), "c2ws": t("all.c2ws")})
    assert model.has_vol and len(model.volumes) == nl and len(model.features) == 5

    def thin(t, i):          # the fixture keeps every second voxel per axis of levels above 32^3
        return t[..., ::2, ::2, ::2] if dims[i] > 32 else t

    for i in range(nl):
        assert rel(thin(model.volumes[i], i), g[f"init.volume{i}"]) < 2e-4, i
        if dims[i] > 32:
            bits = np.packbits(model.mask_volmes[i].cpu().numpy().astype(np.uint8).reshape(-1))
            assert np.array_equal(bits, g[f"init.mask{i}"]), i
        else:
            assert torch.equal(model.mask_volmes[i].cpu(), torch.from_numpy(g[f"init.mask{i}"])), i
        assert model.volumes[i].requires_grad and not model.mask_volmes[i].requires_grad
    for i in range(5):
        assert rel(model.features[i], g[f"init.feature{i}"]) < 2e-4, i
    ipts = {k[3:]: (g[k].tolist() if k == "in.view_ids" else t(k)) for k in g if k.startswith("in.")}
    torch.manual_seed(seed + 3)
    out = model("finetune", ipts, cos_anneal_ratio=1.0, step=11)
    hit = out["mid_inside_sphere"].reshape(1, -1, 1, 1)
    loss = (out["color_fine"].abs().sum() + 0.1 * out["gradient_error"] + 0.01 * out["smooth_error"] + 0.01 * out["tv_reg"]
            + torch.exp(-out["sparse_sdf"].abs() * 100).mean() + (((out["sampled_gray_val"] - out["ref_gray_val"]) ** 2) * hit).mean()
            + 0.1 * out["render_depth"].sum() + out["pseudo_sdf"].abs().mean())
    loss.backward()
    # (free-running hierarchical samples: the inverse-CDF step amplifies the float32 round-off of the weight-normed matrices on rays with a
    # flat pdf; per-sample `weights` feel it first.  The fused step forms g v / |v| with a correctly rounded |v| (float64 sum), the
    # reference's hook with torch._weight_norm's float32 sum: 1.1e-3 here, 0.6e-3 when the step took torch's product itself.  With pinned
    # samples everything agrees to 1e-4: tests/test_hip_render.py)
    bad = {k: e for k, e in ((k, rel(out[k[4:]], v)) for k, v in g.items() if k.startswith("out.")) if e > (2e-3 if k == "out.weights" else 3e-4)}
    assert not bad, bad
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    rows = [(f"volume{i}", rel(thin(model.volumes[i].grad, i), g[f"grad.volume{i}"]), float(np.abs(g[f"grad.volume{i}"]).max())) for i in range(nl)]
    rows.append(("lin0.weight_v", rel(model.implicit_surface.sdf_network.lin0.weight_v.grad, g["grad.lin0"]), float(np.abs(g["grad.lin0"]).max())))
    params = dict(model.named_parameters())
    rows += [(k[5:], rel(params[k[5:]].grad, v), float(np.abs(v).max())) for k, v in g.items() if k.startswith("grad.implicit_surface.")]
    # g18b: the volumes come out of `init_volumes` (device CNNs, within 2e-4 of ATen's) and a 1e-4 perturbation of the volumes alone moves the
    # weight_v gradients of lin1..6 by 1.2e-3 (scripts/probe/oracle_vs_reference_training.py); with the reference's own volumes the CPU oracle
    # reproduces them to 6e-6
    _check_grad_table(rows, loose_weight_v=tag.endswith("l5"))
