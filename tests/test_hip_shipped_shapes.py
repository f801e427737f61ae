"""The reference's SHIPPED shapes at full size (BASELINE configs [3] and [4]), checked against the CPU oracle on a ray sample plus
size-independent properties:

  * the DTU test protocol, confs/gens.conf:17-30: num_src_view = 2 (three views), 480 x 640, the five-level pyramid 256 ... 16 -- a whole
    `validate` image through the fused kernels (gens_blend_views_t at S = 2, gens_sdf_grad<5>);
  * the per-scene fine-tune configuration, confs/gens_finetune.conf:5-16,52-54: img_hw = [1152, 1600], num_views = 3, n_rays = 512,
    volume_dims 256 ... 16 as parameters -- one `forward("finetune")` + backward through K17 / K18 / K8 / K9 / K10 / K13.
The oracle (oracle/render_oracle.py, pinned to the reference by goldens g9a-d / g15 / g17 / g18) renders the same rays with the same
host-generator draws on the CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DIMS5 = [256, 128, 64, 32, 16]


def _surface(seed=0, perturb_weights=0.02, dims=DIMS5):
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.implicit_surface import ImplicitSurface
    torch.manual_seed(seed)
    surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"])
    with torch.no_grad():                                  # off the geometric initialisation, so that the volume features matter
        for p in surf.sdf_network.parameters():
            p.add_(perturb_weights * torch.randn_like(p))
        for p in surf.color_network.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return surf


def _scene(nv, h, w, seed, dims=DIMS5):
    from gens_amd import ops, synthetic
    sc = synthetic.make_scene(nv=nv, h=h, w=w, n_levels=5, seed=seed)
    dev = torch.device("cuda")
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
    d["features"] = [f.to(dev) for f in sc["features"]]
    d["cpu"] = sc
    with torch.no_grad():
        _, d["masks"] = ops.volume_build(d["features"][:len(dims)], d["intrs"], d["c2ws"], dims)
    d["vols_cpu"] = synthetic.make_volumes(dims, seed=seed + 1)
    d["vols"] = [v.to(dev) for v in d["vols_cpu"]]
    return d


@pytest.mark.parametrize("nv,dims", [(3, DIMS5), (5, [256, 128, 64])])
def test_validate_full_image_480x640(nv, dims):
    """The full 307 200-ray image of `validate` against the oracle on a ray sample: BASELINE config[3]'s shape on one GPU (two source views, five
    levels) and config[1] -- the headline workload of bench.py (four source views, volume_dims 256 / 128 / 64)."""
    from gens_amd import synthetic
    from gens_amd.models.modules.implicit_surface import Scene, reference_jitter
    from oracle import render_oracle as R
    sc = _scene(nv, 480, 640, seed=30, dims=dims)
    surf = _surface(1, dims=dims).cuda().eval()
    ro, rd = synthetic.make_rays(sc["cpu"]["intrs"], sc["cpu"]["c2ws"], 480, 640)
    n_rays = ro.shape[0]
    hw = torch.tensor([480, 640]).int()
    scene = Scene(sc["vols"], sc["masks"], sc["imgs"], sc["features"], sc["features"], sc["intrs"], sc["c2ws"])

    def image(chunk):
        surf.val_chunk = chunk
        torch.manual_seed(77)
        with torch.no_grad():
            out = surf.validate(ro.cuda(), rd.cuda(), sc["near"], sc["far"], sc["vols"], sc["masks"], sc["imgs"], sc["features"], sc["features"],
                                sc["intrs"], sc["c2ws"], None, None, hw, extract_geometry=False, scene=scene)
        return out, surf.last_device_image.clone()

    out, dev_image = image(32768)
    assert out["img_fine"].shape == (480, 640, 3) and out["render_depth"].shape == (480, 640)
    assert torch.isfinite(dev_image).all()
    # rays are independent: another chunking renders the same image (same jitter per ray) bit for bit
    _, dev_image2 = image(8192)
    assert torch.equal(dev_image, dev_image2)
    # the oracle on a sample of rays with the SAME jitter (the reference's draw order, chunk by chunk)
    torch.manual_seed(77)
    jitter = reference_jitter(n_rays)
    g = torch.Generator().manual_seed(3)
    pick = torch.randint(0, n_rays, (24,), generator=g)
    sd = {k: v.detach().cpu() for k, v in surf.state_dict().items()}
    masks_c = [m.cpu() for m in sc["masks"]]
    cpu = sc["cpu"]
    ref = R.render(sd, ro[pick], rd[pick], cpu["near"], cpu["far"], sc["vols_cpu"], masks_c, cpu["imgs"], cpu["features"], cpu["features"],
                   cpu["intrs"], cpu["c2ws"], 1.0, None, jitter[pick], torch.rand(1024, 3, generator=g) * 2 - 1)
    got = dev_image[pick.cuda()].cpu()
    # north-star bound: colour / depth L1 within 1e-4 of the reference
    assert (got[:, 0:3] - ref["color_fine"]).abs().mean() < 1e-4
    assert (got[:, 7] - ref["render_depth"].reshape(-1)).abs().mean() < 1e-4
    assert (got[:, 6] - ref["sdf_depth"].reshape(-1)).abs().mean() < 1e-4
    normal = (ref["gradients"] * ref["weights"][..., None] * ref["inside_sphere"][..., None]).sum(1)
    assert (got[:, 3:6] - normal).abs().mean() < 1e-4
    # the opt-in split-half arithmetic (gens_sdf_value_f16 + gens_sdf_grad_f16 at three and at five levels): the same image, same jitter,
    # within a tenth of the north-star bound of the float32 image over ALL rays, and within the bound of the oracle on the sample
    surf.sdf_precision = "f16x2"
    try:
        _, half_image = image(32768)
    finally:
        surf.sdf_precision = "f32"
    assert surf._sdf_plan.grad_pieces is not None and not torch.equal(half_image, dev_image)
    assert (half_image[:, 0:3] - dev_image[:, 0:3]).abs().mean() < 1e-5        # measured 3.4e-6
    assert (half_image[:, 7] - dev_image[:, 7]).abs().mean() < 1e-5            # measured 6.4e-7
    assert (half_image[:, 3:6] - dev_image[:, 3:6]).abs().mean() < 1e-5
    got_h = half_image[pick.cuda()].cpu()
    assert (got_h[:, 0:3] - ref["color_fine"]).abs().mean() < 1e-4 and (got_h[:, 7] - ref["render_depth"].reshape(-1)).abs().mean() < 1e-4


def _finetune_loss(out):
    hit = out["mid_inside_sphere"].reshape(1, -1, 1, 1)
    mfc = (((out["sampled_gray_val"] - out["ref_gray_val"]) ** 2) * hit).mean()
    return (out["color_fine"].abs().sum() + 0.1 * out["gradient_error"] + 0.01 * out["smooth_error"] + 0.01 * out["tv_reg"]
            + torch.exp(-out["sparse_sdf"].abs() * 100).mean() + mfc + 0.1 * out["render_depth"].sum())


def test_finetune_step_three_views_1152x1600_five_levels():
    """BASELINE config[4] shape: the fine-tune step of confs/gens_finetune.conf on one GPU.  The device step runs all 512 rays; the oracle
    (autograd on the CPU, sampler truncated at second order like the reference's Function pair) runs the first 48 of them with the same
    pinned samples, and both sides' gradients are compared on that sub-batch."""
    from gens_amd import synthetic
    from oracle import render_oracle as R
    h, w = 1152, 1600
    sc = _scene(3, h, w, seed=40)
    surf = _surface(2).cuda().train()
    g = torch.Generator().manual_seed(9)
    pix = torch.stack([torch.randint(0, w, (512,), generator=g), torch.randint(0, h, (512,), generator=g)], -1)
    ro, rd = synthetic.make_rays(sc["cpu"]["intrs"], sc["cpu"]["c2ws"], h, w, pixels=pix)
    t_rand = torch.rand(512, 1, generator=g)
    pts_rand = torch.rand(1024, 3, generator=g) * 2 - 1
    vols = [v.clone().requires_grad_(True) for v in sc["vols"]]
    ipts = {"imgs": sc["imgs"], "intrs": sc["intrs"], "c2ws": sc["c2ws"], "rays_o": ro.cuda(), "rays_d": rd.cuda(), "near": sc["near"],
            "far": sc["far"], "pseudo_pts": (torch.rand(2048, 3, generator=g) - 0.5).cuda()}
    # the whole 512-ray step: finite outputs, a gradient for every parameter and every volume level
    torch.manual_seed(5)
    out = surf("finetune", ipts, vols, sc["masks"], sc["features"], sc["features"], 1.0, 11.0)
    assert out["color_fine"].shape == (512, 3) and out["pseudo_sdf"].shape == (2048, 1) and out["ref_gray_val"].shape == (1, 512, 121, 12)
    assert out["sampled_gray_val"].shape == (2, 512, 121, 12)                    # two source views
    (_finetune_loss(out) + out["pseudo_sdf"].abs().mean()).backward()
    for k, p in surf.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    for v in vols:
        assert v.grad is not None and torch.isfinite(v.grad).all() and float(v.grad.abs().sum()) > 0
    # 48 rays against the oracle, samples pinned to the device's own (the inverse-CDF step amplifies round-off, tests/test_hip_render.py)
    nb = 48
    sub = slice(0, nb)
    with torch.no_grad():
        from gens_amd.models.modules.implicit_surface import Scene
        scene = Scene(sc["vols"], sc["masks"], sc["imgs"], sc["features"], sc["features"], sc["intrs"], sc["c2ws"])
        z0 = sc["near"] + (sc["far"] - sc["near"]) * torch.linspace(0, 1, 64).cuda()[None]
        z0 = (z0.expand(nb, 64) + (t_rand[sub].cuda() - 0.5) * 2.0 / 64).contiguous()
        z = surf._sample_rays(ro[sub].cuda().contiguous(), rd[sub].cuda().contiguous(), z0, scene)
    for p in surf.parameters():
        p.grad = None
    vols_d = [v.detach().clone().requires_grad_(True) for v in sc["vols"]]
    out = surf.render_core(ro[sub].cuda().contiguous(), rd[sub].cuda().contiguous(), z, 2.0 / 64, vols_d, sc["masks"], sc["features"], sc["features"],
                           sc["imgs"], sc["intrs"], sc["c2ws"], 1.0, 11.0, pts_random=pts_rand.cuda())
    _finetune_loss(out).backward()
    cpu = sc["cpu"]
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in surf.state_dict().items()}
    vols_c = [v.clone().requires_grad_(True) for v in sc["vols_cpu"]]
    ref = R.render(sd, ro[sub], rd[sub], cpu["near"], cpu["far"], vols_c, [m.cpu() for m in sc["masks"]], cpu["imgs"], cpu["features"],
                   cpu["features"], cpu["intrs"], cpu["c2ws"], 1.0, 11.0, t_rand[sub], pts_rand, truncated=True, z=z.cpu())
    _finetune_loss(ref).backward()
    assert (out["color_fine"].detach().cpu() - ref["color_fine"].detach()).abs().mean() < 1e-4
    assert (out["render_depth"].detach().cpu() - ref["render_depth"].detach()).abs().mean() < 1e-4
    for k in ("gradient_error", "smooth_error", "tv_reg"):
        a, b = float(out[k]), float(ref[k])
        assert abs(a - b) <= 2e-3 * abs(b) + 1e-5, (k, a, b)
    top = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    worst = {}
    for name, p in surf.named_parameters():
        r = sd[name].grad
        worst[name] = float((p.grad.cpu() - r).abs().max()) / max(float(r.abs().max()), 1e-4 * top)
    for i, (a, b) in enumerate(zip(vols_d, vols_c)):
        worst[f"volume{i}"] = float((a.grad.cpu() - b.grad).abs().max()) / max(float(b.grad.abs().max()), 1e-12)
    print({k: f"{v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:8]})
    scalars = {k for k, p in surf.named_parameters() if p.numel() <= 4}          # sums of cancelling per-sample terms: compared loosely
    bad = {k: v for k, v in worst.items() if v >= (0.15 if k in scalars else 2e-3)}
    assert not bad, bad
