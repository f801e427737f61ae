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


def _surface(seed=0, perturb_weights=0.04, dims=DIMS5, radius=0.5):
    """Geometric initialisation (a sphere of `radius`: sdf_network.py:63-88 with bias = radius) moved off it so that the volume features
    matter: every parameter by `perturb_weights` x N(0, 1) x (its mean magnitude + 0.02), the recipe of the render goldens
    (tests/golden/make_golden.py:_perturb).  (Until round 4 the perturbation was ABSOLUTE, 0.02 N(0, 1): that pushed the SDF above +1
    everywhere -- no ray of the full-size tests had a zero crossing, so `sdf_depth` and the surface gate were compared on zeros.)"""
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.implicit_surface import ImplicitSurface
    torch.manual_seed(seed)
    surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"])
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for p in surf.sdf_network.parameters():
            p.add_(perturb_weights * torch.randn(p.shape, generator=g) * (p.abs().mean() + 0.02))
        for p in surf.color_network.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
        last = getattr(surf.sdf_network, f"lin{surf.sdf_network.num_layers - 2}")
        last.bias[0] -= (radius - 0.5)                     # row 0 is the SDF: |x| - radius
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


def ops_coarse_z(sc, surf, t_rand, dev):
    from gens_amd import ops
    return ops.coarse_z(sc["near"], sc["far"], surf._coarse_steps(dev), t_rand.to(dev), t_rand.shape[0])


def _ray_strata(surf, scene, sc, ro, rd, jitter, g, n_cand=24576):
    """Ray indices by category, estimated ON THE DEVICE from the samples validate() will use (same jitter): per-sample SDF through the
    PyTorch layers on K2, the masked first sign change and the unit-sphere gate as render_core applies them (implicit_surface.py:262-281)."""
    from gens_amd import ops
    h, w = 480, 640
    n_rays = ro.shape[0]
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    border = ((xx == 0) | (xx == w - 1) | (yy == 0) | (yy == h - 1)).reshape(-1)
    border_idx = torch.nonzero(border)[:, 0]
    assert ro.shape[0] == h * w
    cand = torch.unique(torch.cat([border_idx, torch.randint(0, n_rays, (n_cand,), generator=g)]))
    dev = torch.device("cuda")
    ro_c, rd_c = ro[cand].to(dev).contiguous(), rd[cand].to(dev).contiguous()
    with torch.no_grad():
        z0 = ops.coarse_z(sc["near"], sc["far"], surf._coarse_steps(dev), jitter[cand].to(dev), cand.numel())
        z = surf._sample_rays(ro_c, rd_c, z0, scene)
        pts, valid = ops.ray_points(ro_c, rd_c, z, scene.masks, mid=True, sample_dist=2.0 / 64)
        sdf = surf.sdf_network.sdf(pts, scene.volumes_nograd()).reshape(z.shape)
        vm = valid.reshape(z.shape).bool()
        sdf = torch.where(vm, sdf, torch.full_like(sdf, 100.0))
        inside = (pts.norm(dim=-1).reshape(z.shape) < 1.0) & vm
        change = (sdf[:, :-1] * sdf[:, 1:] <= 0) & vm[:, :-1] & vm[:, 1:]
        rank = torch.arange(z.shape[1] - 1, 0, -1, device=dev, dtype=torch.float32)
        i0 = torch.argmax(change.float() * rank[None], 1, keepdim=True)
        both_in = inside.gather(1, i0) & inside.gather(1, i0 + 1)
        any_change = change.any(1)
    any_change, both_in = any_change.cpu(), both_in.reshape(-1).cpu()
    return {"border": border_idx, "sphere_gated": cand[any_change & ~both_in], "crossing": cand[any_change & both_in], "no_crossing": cand[~any_change]}


@pytest.mark.parametrize("nv,dims", [(3, DIMS5), (5, [256, 128, 64])])
def test_validate_full_image_480x640(nv, dims):
    """The full 307 200-ray image of `validate` against the oracle on a ray sample: BASELINE config[3]'s shape on one GPU (two source views, five
    levels) and config[1] -- the headline workload of bench.py (four source views, volume_dims 256 / 128 / 64)."""
    from gens_amd import synthetic
    from gens_amd.models.modules.implicit_surface import Scene, reference_jitter
    from oracle import render_oracle as R
    sc = _scene(nv, 480, 640, seed=30, dims=dims)
    surf = _surface(1, dims=dims, radius=0.97 if nv == 5 else 0.6).cuda().eval()      # nv = 5: a surface AT the unit sphere, where render_core's gate decides
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

    out, dev_image = image(None)                                 # the shipped default: validate() cuts the image into equal chunks itself
    assert surf.last_val_chunk == 30720                          # (307 200 rays = 10 x 30 720: what bench.py's headline runs)
    assert out["img_fine"].shape == (480, 640, 3) and out["render_depth"].shape == (480, 640)
    assert torch.isfinite(dev_image).all()
    # rays are independent: another chunking renders the same image (same jitter per ray) bit for bit
    _, dev_image2 = image(8192)
    assert torch.equal(dev_image, dev_image2)
    # the oracle on a STRATIFIED sample of rays with the SAME jitter (the reference's draw order, chunk by chunk): image-border pixels, rays
    # whose first masked sign change is rejected by the unit-sphere gate (implicit_surface.py:277-281), rays with a crossing that passes, rays
    # without any crossing, plus a uniform draw -- 512 rays for either shape
    torch.manual_seed(77)
    jitter = reference_jitter(n_rays)
    g = torch.Generator().manual_seed(3)
    n_oracle = 512                      # (round 4: 128 for the five-level protocol; the GPU box's 128 host cores do 640 oracle rays in 12 s)
    strata = _ray_strata(surf, scene, sc, ro, rd, jitter, g)
    quota = {"border": n_oracle // 4, "sphere_gated": n_oracle // 8, "crossing": n_oracle // 8, "no_crossing": n_oracle // 8}
    chosen, taken = [], {}
    for name, share in quota.items():
        cand = strata[name]
        take = cand[torch.randperm(cand.numel(), generator=g)[:share]]
        taken[name] = int(take.numel())
        chosen.append(take)
    chosen.append(torch.randint(0, n_rays, (n_oracle - sum(taken.values()),), generator=g))
    pick = torch.cat(chosen)
    print("oracle rays per stratum:", taken, "+ uniform", int(chosen[-1].numel()), "| candidates:", {k: int(v.numel()) for k, v in strata.items()})
    assert taken["border"] == quota["border"] and taken["crossing"] > 0
    sd = {k: v.detach().cpu() for k, v in surf.state_dict().items()}
    masks_c = [m.cpu() for m in sc["masks"]]
    cpu = sc["cpu"]
    ref = R.render(sd, ro[pick], rd[pick], cpu["near"], cpu["far"], sc["vols_cpu"], masks_c, cpu["imgs"], cpu["features"], cpu["features"],
                   cpu["intrs"], cpu["c2ws"], 1.0, None, jitter[pick], torch.rand(1024, 3, generator=g) * 2 - 1)
    got = dev_image[pick.cuda()].cpu()
    # what the ORACLE says the sample contains (the strata were drawn from a device-side estimate): gated crossings, accepted ones, none
    n_ = ref["weights"].shape[1]
    sdf_o = ref["sparse_sdf"][1024:].reshape(-1, n_)
    accepted = ref["mid_inside_sphere"].reshape(-1) > 0
    has_change = ((sdf_o[:, :-1] * sdf_o[:, 1:] <= 0) & (sdf_o[:, :-1] < 99) & (sdf_o[:, 1:] < 99)).any(1)
    print("oracle: accepted crossings %d, sign change but rejected %d, no sign change %d, rays failing the >8-visible-samples test %d"
          % (int(accepted.sum()), int((has_change & ~accepted).sum()), int((~has_change).sum()), int((~ref["valid_mask"].reshape(-1)).sum())))
    assert int(accepted.sum()) >= n_oracle // 16 and int((~accepted).sum()) >= n_oracle // 16
    # The same rays once more with the hierarchical samples PINNED to the device's: the reference's inverse-CDF step is discontinuous where a
    # CDF denominator sits at its 1e-5 threshold (implicit_surface.py:36-41) and amplifies 1e-7 of SDF round-off into samples that move by
    # up to a bin, so a handful of rays legitimately render from other samples; with pinned samples everything downstream must agree tightly.
    dev = torch.device("cuda")
    with torch.no_grad():
        ro_p, rd_p = ro[pick].to(dev).contiguous(), rd[pick].to(dev).contiguous()
        z_dev = surf._sample_rays(ro_p, rd_p, ops_coarse_z(sc, surf, jitter[pick], dev), scene)
        again = surf.render_core(ro_p, rd_p, z_dev, 2.0 / 64, sc["vols"], sc["masks"], sc["features"], sc["features"], sc["imgs"], sc["intrs"],
                                 sc["c2ws"], 1.0, None, scene=scene, lean=True)
    assert torch.equal(again["color_fine"], dev_image[pick.cuda(), 0:3]) and torch.equal(again["render_depth"].reshape(-1), dev_image[pick.cuda(), 7])
    pinned = R.render(sd, ro[pick], rd[pick], cpu["near"], cpu["far"], sc["vols_cpu"], masks_c, cpu["imgs"], cpu["features"], cpu["features"],
                      cpu["intrs"], cpu["c2ws"], 1.0, None, jitter[pick], torch.rand(1024, 3, generator=g) * 2 - 1, z=z_dev.cpu())
    z_ref = R.sample_rays(sd, ro[pick], rd[pick], cpu["near"], cpu["far"], sc["vols_cpu"], masks_c, jitter[pick])
    dz = (z_dev.cpu() - z_ref).abs().max(1)[0]
    print("hierarchical samples, worst |device - oracle| per ray: median %.1e, p99 %.1e, max %.1e; rays with a sample off by > 1e-4: %d of %d"
          % (float(dz.median()), float(dz.quantile(0.99)), float(dz.max()), int((dz > 1e-4).sum()), dz.numel()))
    # the discontinuity is rare, the rest is round-off.  Observed (profiles/r06_parity_numbers.txt): 1 of 512 rays at three levels / four source views,
    # 8 of 512 at five levels / two source views (border rays most); the bound is twice the worst observed count (round 5: n / 16 = 32)
    assert float(dz.median()) <= 2e-6 and int((dz > 1e-4).sum()) <= max(2, n_oracle // 32)

    def per_ray(r):
        nrm = (r["gradients"] * r["weights"][..., None] * r["inside_sphere"][..., None]).sum(1)
        return {"colour": (got[:, 0:3] - r["color_fine"]).abs().mean(1), "render_depth": (got[:, 7] - r["render_depth"].reshape(-1)).abs(),
                "sdf_depth": (got[:, 6] - r["sdf_depth"].reshape(-1)).abs(), "normal": (got[:, 3:6] - nrm).abs().mean(1)}
    free_err, pin_err = per_ray(ref), per_ray(pinned)
    bounds = [0] + torch.cumsum(torch.tensor([c.numel() for c in chosen]), 0).tolist()
    failures = []
    for name, a, b in [("all", 0, pick.numel())] + [(nm, bounds[i], bounds[i + 1]) for i, nm in enumerate(list(quota) + ["uniform"])]:
        if b == a:
            continue
        for label, err in (("oracle's samples", free_err), ("pinned samples", pin_err)):
            print("  %-12s %4d rays  %-16s " % (name, b - a, label)
                  + "  ".join("%s mean %.1e max %.1e" % (k, float(v[a:b].mean()), float(v[a:b].max())) for k, v in err.items()))
        # pinned samples: the north-star bound (colour / depth L1 within 1e-4) holds inside EVERY stratum, with a tenth of it as the typical ray
        for k, v in pin_err.items():
            if float(v[a:b].mean()) >= 1e-4 or float(v[a:b].median()) >= 1e-5:
                failures.append((name, "pinned", k, float(v[a:b].mean()), float(v[a:b].median())))
        # the oracle's own samples: the bound holds on the uniform draw (= the image's L1) for colour and both depths
        if name == "uniform":
            for k in ("colour", "render_depth", "sdf_depth"):
                if float(free_err[k][a:b].mean()) >= 1e-4:
                    failures.append((name, "free", k, float(free_err[k][a:b].mean())))
    assert not failures, failures
    # the accept / reject decision of the zero-crossing depth itself, ray by ray: a ray may flip only if its crossing sits ON a gate
    flipped = ((got[:, 6] != 0) != (ref["sdf_depth"].reshape(-1) != 0))
    assert int(flipped.sum()) <= max(1, n_oracle // 128), int(flipped.sum())

    # the opt-in split-half arithmetic (gens_sdf_value_f16 + gens_sdf_grad_f16 at three and at five levels): the same image, same jitter,
    # within a tenth of the north-star bound of the float32 image over ALL rays, and within the bound of the oracle on the sample
    surf.sdf_precision = "f16x2"
    try:
        _, half_image = image(None)
    finally:
        surf.sdf_precision = "f32"
    assert surf._sdf_plan.grad_pieces is not None and not torch.equal(half_image, dev_image)
    # (half the north-star bound.  Measured on these scenes, which have a surface: colour 1.6e-5 at five levels / two source views, below 1e-5 at
    # three levels; the typical ray moves by 1e-6 -- the mean is carried by the rays whose hierarchical samples jump at the inverse-CDF
    # threshold, the same discontinuity that separates two float32 implementations above)
    print("split-half vs float32 image: colour %.1e, depth %.1e, normal %.1e (mean |diff| over all rays); median colour %.1e"
          % (float((half_image[:, 0:3] - dev_image[:, 0:3]).abs().mean()), float((half_image[:, 7] - dev_image[:, 7]).abs().mean()),
             float((half_image[:, 3:6] - dev_image[:, 3:6]).abs().mean()), float((half_image[:, 0:3] - dev_image[:, 0:3]).abs().mean(1).median())))
    assert (half_image[:, 0:3] - dev_image[:, 0:3]).abs().mean() < 5e-5
    assert (half_image[:, 7] - dev_image[:, 7]).abs().mean() < 5e-5
    assert (half_image[:, 0:3] - dev_image[:, 0:3]).abs().mean(1).median() < 5e-6
    got_h = half_image[pick.cuda()].cpu()
    u = slice(bounds[-2], bounds[-1])                     # the uniform draw (= the image's L1; the border stratum over-weights the sampling discontinuity)
    assert (got_h[u, 0:3] - ref["color_fine"][u]).abs().mean() < 1e-4 and (got_h[u, 7] - ref["render_depth"].reshape(-1)[u]).abs().mean() < 1e-4


class _f64:
    """Run the oracle in float64 (its tensors follow the default dtype): the yardstick that tells float32 round-off from a wrong formula."""

    def __enter__(self):
        torch.set_default_dtype(torch.float64)

    def __exit__(self, *exc):
        torch.set_default_dtype(torch.float32)


def _to64(t):
    if isinstance(t, (list, tuple)):
        return [_to64(x) for x in t]
    return t.double() if torch.is_tensor(t) and t.is_floating_point() else t


def _judge_gradients(rows, tol=2e-3):
    """rows: {name: (device gradient, float32-oracle gradient, float64-oracle gradient)}.  A device gradient passes if it is within `tol` of
    the float32 oracle relative to that tensor's largest entry -- or, where float32 itself cannot do better (sums of cancelling second-order
    terms), if it is as close to the FLOAT64 oracle as the float32 oracle is (3 x: two float32 implementations that associate differently).
    Prints the table; -> {name: what failed}."""
    bad = {}
    print("%-44s %10s %12s %12s" % ("gradient", "dev-o32", "dev-o64", "o32-o64   (max |diff| / max |o64|)"))
    # a network parameter whose gradient is zero by symmetry (the bias in front of a softmax) is pure round-off on every side: its errors are
    # measured against 1e-4 of the largest parameter gradient of the step, as tests/test_hip_training.py does
    top = max(float(o64.abs().max()) for name, (_, _, o64) in rows.items() if "network" in name)
    for name, (dev_g, o32, o64) in sorted(rows.items()):
        dev_g, o32, o64 = dev_g.detach().double().cpu(), o32.detach().double(), o64.detach().double()
        scale = max(float(o64.abs().max()), 1e-4 * top if "network" in name else 1e-30)
        e_do, e_d64, e_o = float((dev_g - o32).abs().max()) / scale, float((dev_g - o64).abs().max()) / scale, float((o32 - o64).abs().max()) / scale
        ok = e_do < tol or e_d64 <= 3.0 * e_o
        if not ok or e_do >= tol:
            print("%-44s %10.1e %12.1e %12.1e %s" % (name, e_do, e_d64, e_o, "" if ok else "  <-- FAIL"))
        if not ok:
            bad[name] = (e_do, e_d64, e_o)
            worst = torch.topk((dev_g - o64).abs().reshape(-1), min(5, dev_g.numel())).indices      # where: one entry (a decision that fell the other way) or everywhere (round-off)?
            print("    worst entries of %s (flat index: device / float32 oracle / float64 oracle): %s; entries beyond tol * scale: %d of %d" % (
                name, [(int(i), float(dev_g.reshape(-1)[i]), float(o32.reshape(-1)[i]), float(o64.reshape(-1)[i])) for i in worst],
                int(((dev_g - o64).abs() > tol * scale).sum()), dev_g.numel()))
    return bad


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
    # 128 rays against the oracle, samples pinned to the device's own (the inverse-CDF step amplifies round-off, tests/test_hip_render.py)
    nb = 128
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
    masks_c = [m.cpu() for m in sc["masks"]]

    def oracle(cast):
        sd_ = {k: cast(v.detach().cpu()).clone().requires_grad_(True) for k, v in surf.state_dict().items()}
        vols_ = [cast(v).clone().requires_grad_(True) for v in sc["vols_cpu"]]
        r = R.render(sd_, cast(ro[sub]), cast(rd[sub]), cast(cpu["near"]), cast(cpu["far"]), vols_, cast(masks_c), cast(cpu["imgs"]), cast(cpu["features"]),
                     cast(cpu["features"]), cast(cpu["intrs"]), cast(cpu["c2ws"]), 1.0, 11.0, cast(t_rand[sub]), cast(pts_rand), truncated=True, z=cast(z.cpu()))
        _finetune_loss(r).backward()
        return r, sd_, vols_
    ref, sd, vols_c = oracle(lambda t: t)
    with _f64():
        _, sd64, vols64 = oracle(_to64)
    _assert_training_strata(ref, nb)
    assert (out["color_fine"].detach().cpu() - ref["color_fine"].detach()).abs().mean() < 1e-4
    assert (out["render_depth"].detach().cpu() - ref["render_depth"].detach()).abs().mean() < 1e-4
    for k in ("gradient_error", "smooth_error", "tv_reg"):
        a, b = float(out[k]), float(ref[k])
        assert abs(a - b) <= 2e-3 * abs(b) + 1e-5, (k, a, b)
    rows = {name: (p.grad, sd[name].grad, sd64[name].grad) for name, p in surf.named_parameters()}
    rows.update({f"volume{i}": (a.grad, b.grad, c.grad) for i, (a, b, c) in enumerate(zip(vols_d, vols_c, vols64))})
    bad = _judge_gradients(rows)
    assert not bad, bad



def _assert_training_strata(ref, nb):
    """What a training sub-batch must contain for its comparison to mean something (round 4's lesson: a stratum that is empty is compared on
    zeros): rays the losses count (valid_mask), rays whose first sign change passes the sphere gate and rays where it does not, and
    compositing weights that are not all ~0."""
    valid = ref["valid_mask"].reshape(-1).bool()
    mid = ref["mid_inside_sphere"].reshape(-1) > 0.5
    assert int(valid.sum()) >= nb // 4, int(valid.sum())
    assert int(mid.sum()) >= nb // 16 and int((~mid).sum()) >= nb // 16, (int(mid.sum()), int((~mid).sum()))
    assert float(ref["weight_sum"].detach().max()) > 0.5

def _band_limited(features):
    """Feature maps with period >= 40 px (tests/test_hip_kernels.py::test_k1_volume_vs_oracle_480x640: white noise turns the 1e-4 px float32
    uncertainty of a projected coordinate into 1e-3 value differences between ANY two implementations)."""
    out = []
    for i, f in enumerate(features):
        nv, c, h, w = f.shape
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        ph = torch.arange(nv * c, dtype=torch.float32).reshape(nv, c, 1, 1)
        out.append(torch.sin(xx * (0.15 * 2 ** i / (1 + ph % 3)) + yy * (0.11 * 2 ** i) + ph) * (1 + 0.1 * ph))
    return out


def _rel_off(a, b, tol, floor):
    """Fraction of elements of a that differ from b by more than tol * max(|b|.max(), floor)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float(((a - b).abs() > tol * max(float(b.abs().max()), floor)).float().mean()), float((a - b).abs().max())


def test_training_step_config2_shape_five_views_480x640_volumes_256_128_64():
    """BASELINE config[2] at its OWN size: 5 views 480 x 640, volume_dims 256 / 128 / 64, 512 rays + 2048 pseudo points, the reference's Loss,
    backward through every kernel -- with the K1 build inside the differentiated graph.

      leg 1  K1 forward AND backward at 256^3 / 128^3 / 64^3 against the oracle on voxel slabs: the cotangent lives on a 8 x 256 x 256 slab of
             level 0, a 16 x 128 x 128 slab of level 1 and the whole of level 2, so the CPU builds those voxels only (volume.py:27-61: voxels are
             independent; the oracle's slab form is pinned on the golden sizes by tests/test_oracle_golden.py).
      leg 2  the whole 512-ray step (GenS.forward's render + Loss + backward, K1 backward in the same graph): every parameter, every volume
             level and every feature level receives a finite gradient.
      leg 3  a 32-ray sub-batch of the same step against the oracle: render (samples pinned to the device's), oracle/loss_oracle.py
             (= loss.py:23-84), backward into the MLPs, the volumes and the feature pyramid (K4's scatter + K1's slab term)."""
    from gens_amd import ops, synthetic
    from gens_amd.config import gens_loss_conf
    from gens_amd.losses import Loss
    from gens_amd.models.modules.implicit_surface import Scene
    from oracle import gens_oracle as K
    from oracle import loss_oracle
    from oracle import render_oracle as R
    dims = [256, 128, 64]
    h, w = 480, 640
    dev = torch.device("cuda")
    cpu = synthetic.make_scene(nv=5, h=h, w=w, n_levels=5, seed=50)
    cpu["features"] = _band_limited(cpu["features"])
    intrs, c2ws, imgs = cpu["intrs"].to(dev), cpu["c2ws"].to(dev), cpu["imgs"].to(dev)
    g = torch.Generator().manual_seed(11)
    ranges = [(124, 132), (40, 56), (0, 64)]
    cots_c = []
    for d, (x0, x1) in zip(dims, ranges):
        c = torch.zeros(1, 8, d, d, d)
        c[:, :, x0:x1] = torch.randn(1, 8, x1 - x0, d, d, generator=g)
        cots_c.append(c)
    cots = [c.to(dev) for c in cots_c]

    # ---- leg 1: K1 at full size against the oracle on the slabs
    feats = [f.to(dev).requires_grad_(True) for f in cpu["features"]]
    cost, masks = ops.volume_build(feats[:3], intrs, c2ws, dims)
    g_k1 = torch.autograd.grad(sum((a * b).sum() for a, b in zip(cost, cots)), feats[:3])
    feats_c = [f.clone().requires_grad_(True) for f in cpu["features"]]
    cost_o, masks_o = K.volume_build(feats_c[:3], cpu["intrs"], cpu["c2ws"], dims, x_ranges=ranges)
    sum((a * c[:, :, x0:x1]).sum() for a, c, (x0, x1) in zip(cost_o, cots_c, ranges)).backward()
    for i, (x0, x1) in enumerate(ranges):
        got_m, got_v = masks[i][:, :, x0:x1].cpu(), cost[i][:, :, x0:x1].detach().cpu()
        # a voxel projecting within an ulp of an image border may flip visibility in one view
        assert float((got_m != masks_o[i]).float().mean()) <= 1e-4, ("mask", i)
        bad = (got_v - cost_o[i].detach()).abs() > 5e-5 + 1e-4 * cost_o[i].detach().abs()
        assert float(bad.float().mean()) <= 1e-4, ("cost volume", i, int(bad.sum()))
        off, worst = _rel_off(g_k1[i], feats_c[i].grad, 2e-4, 1e-6)
        print("K1 backward level %d (D = %d, slab %d..%d): worst |diff| %.2e of max %.2e, %.2e of the texels beyond 2e-4 of the max"
              % (i, dims[i], x0, x1, worst, float(feats_c[i].grad.abs().max()), off))
        assert off <= 1e-4 and worst <= 5e-3 * float(feats_c[i].grad.abs().max()), (i, off, worst)
    k1_grads_c = [f.grad.clone() for f in feats_c[:3]]
    del cost, g_k1

    # ---- leg 2: the whole 512-ray step with the K1 build in the graph
    surf = _surface(3, dims=dims).cuda().train()
    vols_cpu = synthetic.make_volumes(dims, seed=51)
    pix = torch.stack([torch.randint(0, w, (512,), generator=g), torch.randint(0, h, (512,), generator=g)], -1)
    ro, rd = synthetic.make_rays(cpu["intrs"], cpu["c2ws"], h, w, pixels=pix)
    pseudo = torch.rand(2048, 3, generator=g) - 0.5
    near, far = cpu["near"].to(dev), cpu["far"].to(dev)
    ipts = {"imgs": imgs, "intrs": intrs, "c2ws": c2ws, "rays_o": ro.to(dev), "rays_d": rd.to(dev), "near": near, "far": far, "pseudo_pts": pseudo.to(dev)}
    pseudo_depth = torch.where(torch.rand(512, generator=g) < 0.3, torch.zeros(512), 1.0 + 2.0 * torch.rand(512, generator=g))
    targets_c = {"color": torch.rand(512, 3, generator=g), "pseudo_depth": pseudo_depth}
    targets = {k: v.to(dev) for k, v in targets_c.items()}
    loss_fn = Loss(gens_loss_conf()).to(dev)
    vols = [v.to(dev).requires_grad_(True) for v in vols_cpu]
    for f in feats:
        f.grad = None
    torch.manual_seed(5)
    cost, masks = ops.volume_build(feats[:3], intrs, c2ws, dims)
    out = surf("train", ipts, vols, masks, feats, [f.detach() for f in feats], 0.5, 1.0)
    terms = loss_fn(out, targets)
    assert out["color_fine"].shape == (512, 3) and out["pseudo_sdf"].shape == (2048, 1) and out["sampled_gray_val"].shape == (4, 512, 121, 12)
    (terms["loss"] + sum((a * b).sum() for a, b in zip(cost, cots))).backward()
    surf.check_deferred()
    for k, p in surf.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    for v in vols:
        assert v.grad is not None and torch.isfinite(v.grad).all() and float(v.grad.abs().sum()) > 0
    for i, f in enumerate(feats):
        assert f.grad is not None and torch.isfinite(f.grad).all() and float(f.grad.abs().sum()) > 0, i
    del out, terms, cost

    # ---- leg 3: a 96-ray sub-batch of that step against the oracle (samples pinned to the device's)
    nb = 96
    t_rand_all = torch.rand(nb, 1, generator=g)
    pts_rand = torch.rand(1024, 3, generator=g) * 2 - 1
    masks_c = [m.cpu() for m in masks]
    ok_c = K.point_valid(masks_c, pseudo).reshape(-1)
    pseudo_ok = pseudo[ok_c]
    assert 0 < pseudo_ok.shape[0] < 2048 or pseudo_ok.shape[0] == 2048
    scene = Scene([v.detach() for v in vols], masks, imgs, [f.detach() for f in feats], [f.detach() for f in feats], intrs, c2ws)

    def device_step(sub):
        n_sub = sub.shape[0]
        with torch.no_grad():
            z0 = ops.coarse_z(near, far, surf._coarse_steps(dev), t_rand_all[sub].to(dev), n_sub)
            z = surf._sample_rays(ro[sub].to(dev).contiguous(), rd[sub].to(dev).contiguous(), z0, scene)
        for p in surf.parameters():
            p.grad = None
        vols_d = [v.detach().clone().requires_grad_(True) for v in vols]
        feats_d = [f.detach().clone().requires_grad_(True) for f in feats]
        cost, masks_d = ops.volume_build(feats_d[:3], intrs, c2ws, dims)
        flags = torch.empty(2048, device=dev, dtype=torch.uint8)
        ops.lookup_mask(pseudo.to(dev), ops.VolumeSet.masks(masks_d), out=flags)          # as ImplicitSurface.forward hands the pseudo points over
        assert torch.equal(flags.cpu().bool(), ok_c)
        out = surf.render_core(ro[sub].to(dev).contiguous(), rd[sub].to(dev).contiguous(), z, 2.0 / 64, vols_d, masks_d, feats_d,
                               [f.detach() for f in feats_d], imgs, intrs, c2ws, 0.5, 1.0, pts_random=pts_rand.to(dev), extra_pts=pseudo.to(dev),
                               extra_valid=flags)
        out["pseudo_sdf"] = out.pop("_extra_sdf_dense")
        terms = loss_fn(out, {k: v[sub.to(dev)] for k, v in targets.items()})
        (terms["loss"] + sum((a * b).sum() for a, b in zip(cost, cots))).backward()
        return z, vols_d, feats_d, out, terms

    # The reference's sparse term is exp(-100 |sdf|) (loss.py:30): |.| has a kink at the surface, and the hierarchical sampler puts samples ON it --
    # a sample whose SDF is within float32 round-off of zero gets the opposite sign(sdf) on the device and in the oracle (both right), i.e. a
    # gradient contribution of 2 x 100 x weight / N with either sign (found at 96 rays: one such sample moved one level-2 voxel by 4e-6 of a 2e-3
    # tolerance).  Rays that carry such a sample are left out of the comparison -- decided on the device's own values, before the oracle runs.
    sub = torch.arange(nb)
    z, vols_d, feats_d, out, terms = device_step(sub)
    ray_sdf = out["sparse_sdf"].detach()[1024:, 0].reshape(nb, -1)
    on_kink = (ray_sdf.abs() < 1e-5).any(dim=1).cpu()
    if bool(on_kink.any()):
        print("rays with a sample within 1e-5 of the surface (left out):", torch.nonzero(on_kink)[:, 0].tolist())
        assert int(on_kink.sum()) <= nb // 8
        sub = sub[~on_kink]
        z, vols_d, feats_d, out, terms = device_step(sub)
    nb = sub.shape[0]
    t_rand = t_rand_all[sub]
    sub_targets = {k: v[sub.to(dev)] for k, v in targets.items()}

    conf = gens_loss_conf()
    names = ("color_weight", "igr_weight", "sparse_weight", "mfc_weight", "smooth_weight", "tv_weight", "pseudo_sdf_weight", "pseudo_depth_weight",
             "sparse_scale_factor")
    weights = {k: conf.get_float(k) for k in names}

    def oracle(cast):
        sd_ = {k: cast(v.detach().cpu()).clone().requires_grad_(True) for k, v in surf.state_dict().items()}
        vols_ = [cast(v).clone().requires_grad_(True) for v in vols_cpu]
        feats_ = [cast(f.detach()).clone().requires_grad_(True) for f in cpu["features"]]
        r = R.render(sd_, cast(ro[sub]), cast(rd[sub]), cast(cpu["near"]), cast(cpu["far"]), vols_, cast(masks_c), cast(cpu["imgs"]), feats_,
                     [f.detach() for f in feats_], cast(cpu["intrs"]), cast(cpu["c2ws"]), 0.5, 1.0, cast(t_rand), cast(pts_rand), truncated=True, z=cast(z.cpu()))
        ps = torch.zeros(2048, 1, dtype=r["color_fine"].dtype)
        ps[ok_c] = R.sdf_mlp(sd_, cast(pseudo_ok), vols_, lookup=K.lookup_volume_truncated)[:, :1]
        r["pseudo_sdf"] = ps
        terms_ = loss_oracle.loss(r, {k: cast(v[sub]) for k, v in targets_c.items()}, weights)
        terms_["loss"].backward()
        return r, terms_, sd_, vols_, feats_
    ref, ref_terms, sd, vols_c, feats_o = oracle(lambda t: t)
    with _f64():
        _, _, sd64, vols64, feats64 = oracle(_to64)
    _assert_training_strata(ref, nb)
    for k in loss_oracle.TERMS:
        a, b = float(terms[k]), float(ref_terms[k])
        assert abs(a - b) <= 2e-3 * abs(b) + 1e-5, (k, a, b)
    assert (out["color_fine"].detach().cpu() - ref["color_fine"].detach()).abs().mean() < 1e-4
    assert (out["render_depth"].detach().cpu() - ref["render_depth"].detach()).abs().mean() < 1e-4
    rows = {name: (p.grad, sd[name].grad, sd64[name].grad) for name, p in surf.named_parameters()}
    rows.update({f"volume{i}": (a.grad, b.grad, c.grad) for i, (a, b, c) in enumerate(zip(vols_d, vols_c, vols64))})
    for i, (a, b, c) in enumerate(zip(feats_d, feats_o, feats64)):
        # the render's share (K4's scatter) + K1's slab term (leg 1's oracle gradient; the same tensor on both oracle sides)
        k1 = k1_grads_c[i] if i < 3 else torch.zeros_like(b.grad)
        rows[f"feature{i}"] = (a.grad, b.grad + k1, c.grad + k1.double())
    bad = _judge_gradients(rows)
    assert not bad, bad
