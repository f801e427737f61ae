"""GPU parity of the host-side mirror (ImplicitSurface / GenS on the HIP kernels) against the reference's end-to-end
golden vectors and the CPU oracle.  North-star tolerance: depth / colour L1 within 1e-4 of the reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def close(a, b, atol=1e-5, rtol=1e-5, what="", frac=0.0):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if a.numel() == 0:
        return
    bad = (a - b).abs() > atol + rtol * b.abs()
    assert bad.float().mean().item() <= frac, f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {(a - b).abs().max().item():.3e}"


def n_levels(g):
    return sum(1 for k in g if k.startswith("vol"))


def build_surface(g):
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.implicit_surface import ImplicitSurface
    surf = ImplicitSurface(gens_model_conf(volume_dims=(24, 16, 8, 6, 4)[:n_levels(g)])["implicit_surface"])
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    surf.load_state_dict(sd, strict=True)
    return surf.cuda()


def scene_inputs(g):
    c = lambda t: t.cuda()  # noqa: E731
    feats = [c(g[f"feat{i}"]) for i in range(5)]
    vols = [c(g[f"vol{i}"]) for i in range(n_levels(g))]
    masks = [c(g[f"mask{i}"]) for i in range(n_levels(g))]
    match = [f + 0.01 for f in feats]
    step = None if float(g["step"]) < 0 else float(g["step"])
    return feats, vols, masks, match, step


RENDER_GOLDENS = ["g9a_render", "g9b_render", "g9c_render_l5"]      # g9c: five volume levels (the shipped count), four source views


@pytest.mark.parametrize("tag", RENDER_GOLDENS)
def test_render_matches_reference_golden(golden, tag):
    g = golden(tag)
    surf = build_surface(g)
    feats, vols, masks, match, step = scene_inputs(g)
    c = lambda t: t.cuda()  # noqa: E731
    torch.manual_seed(int(g["rng_seed"]))          # same CPU generator state as the reference run
    out = surf.render(c(g["rays_o"]), c(g["rays_d"]), c(g["near"]), c(g["far"]), vols, masks, c(g["imgs"]), feats, match, c(g["intrs"]),
                      c(g["c2ws"]), float(g["cos_anneal"]), step)
    keys = sorted(k[4:] for k in g if k.startswith("out."))
    assert sorted(out.keys()) == keys
    for k in keys:
        assert tuple(out[k].shape) == tuple(g["out." + k].shape), k
    assert (out["color_fine"].cpu() - g["out.color_fine"]).abs().mean() < 1e-4
    assert (out["render_depth"].cpu() - g["out.render_depth"]).abs().mean() < 1e-4
    assert (out["sdf_depth"].cpu() - g["out.sdf_depth"]).abs().mean() < 1e-4
    assert torch.equal(out["valid_mask"].cpu(), g["out.valid_mask"])


@pytest.mark.parametrize("tag", RENDER_GOLDENS)
def test_render_core_matches_reference_golden_with_pinned_samples(golden, tag):
    """Same, but with the reference's hierarchical samples injected, so every output can be compared tightly
    (inverse-CDF sampling amplifies float32 round-off on rays with a flat pdf, see tests/test_oracle_golden.py)."""
    g = golden(tag)
    surf = build_surface(g)
    feats, vols, masks, match, step = scene_inputs(g)
    c = lambda t: t.cuda()  # noqa: E731
    out = surf.render_core(c(g["rays_o"]), c(g["rays_d"]), c(g["z_final"]), 2.0 / 64, vols, masks, feats, match, c(g["imgs"]), c(g["intrs"]),
                           c(g["c2ws"]), float(g["cos_anneal"]), step, pts_random=c(g["draw_ptsrand"]) * 2 - 1)
    assert (out["color_fine"].cpu() - g["out.color_fine"]).abs().mean() < 1e-4
    assert (out["render_depth"].cpu() - g["out.render_depth"]).abs().mean() < 1e-4
    assert torch.equal(out["valid_mask"].cpu(), g["out.valid_mask"])
    close(out["mid_inside_sphere"], g["out.mid_inside_sphere"], atol=0, rtol=0, what="mid_inside_sphere")
    close(out["inside_sphere"], g["out.inside_sphere"], atol=0, rtol=0, what="inside_sphere")
    for k in ["weights", "weight_sum", "weight_max", "normal", "s_val", "sdf_depth", "color_fine", "render_depth"]:
        close(out[k], g["out." + k], atol=1e-4, rtol=1e-3, what=k)
    close(out["gradients"], g["out.gradients"], atol=5e-4, rtol=1e-3, what="gradients")
    close(out["sparse_sdf"], g["out.sparse_sdf"], atol=1e-4, rtol=1e-4, what="sparse_sdf")
    for k in ["gradient_error", "smooth_error", "tv_reg"]:
        close(out[k], g["out." + k], atol=1e-4, rtol=2e-3, what=k)
    hit = g["out.mid_inside_sphere"][:, 0] > 0
    close(out["ref_gray_val"][:, hit], g["out.ref_gray_val"][:, hit], atol=2e-3, rtol=1e-3, what="ref_gray_val")
    close(out["sampled_gray_val"][:, hit], g["out.sampled_gray_val"][:, hit], atol=5e-3, rtol=1e-2, what="sampled_gray_val")


def test_hierarchical_samples_match_reference(golden):
    g = golden("g9a_render")
    surf = build_surface(g)
    feats, vols, masks, match, _ = scene_inputs(g)
    from gens_amd.models.modules.implicit_surface import Scene
    c = lambda t: t.cuda()  # noqa: E731
    scene = Scene(vols, masks, c(g["imgs"]), feats, match, c(g["intrs"]), c(g["c2ws"]))
    z0 = c(g["near"]) + (c(g["far"]) - c(g["near"])) * torch.linspace(0, 1, 64).cuda()[None]
    z0 = (z0.expand(g["rays_o"].shape[0], 64) + (c(g["draw_trand"]) - 0.5) * 2.0 / 64).contiguous()
    z = surf._sample_rays(c(g["rays_o"]), c(g["rays_d"]), z0, scene).cpu()
    err = (z - g["z_final"]).abs()
    assert err.max() < 2e-3 and (err > 1e-4).float().mean() < 0.01, f"max {err.max():.2e}"
    assert (z[:, 1:] >= z[:, :-1]).all()


def test_lean_validation_path_equals_full_render(golden):
    """validate() skips work whose results it discards; the kept outputs must not change."""
    g = golden("g9b_render")
    surf = build_surface(g)
    feats, vols, masks, match, step = scene_inputs(g)
    c = lambda t: t.cuda()  # noqa: E731
    args = (c(g["rays_o"]), c(g["rays_d"]), c(g["z_final"]), 2.0 / 64, vols, masks, feats, match, c(g["imgs"]), c(g["intrs"]), c(g["c2ws"]), 1.0, step)
    surf.fused_train = False                      # `full` on the PyTorch layers, so that the last comparison below is like for like
    full = surf.render_core(*args)
    with torch.no_grad():
        lean = surf.render_core(*args, lean=True)
    # lean runs the fused MFMA SDF kernel, full the PyTorch layers: same float32 arithmetic, different summation order
    for k in ["color_fine", "render_depth", "sdf_depth", "weights", "inside_sphere"]:
        close(lean[k], full[k], atol=2e-5, rtol=1e-4, what=k)
    close(lean["gradients"], full["gradients"], atol=2e-4, rtol=1e-3, what="gradients")
    surf.fused_sdf = surf.fused_blend = False
    with torch.no_grad():
        lean_torch = surf.render_core(*args, lean=True)
    for k in ["color_fine", "render_depth", "sdf_depth", "weights", "gradients", "inside_sphere"]:
        close(lean_torch[k], full[k], atol=1e-6, rtol=1e-5, what=k + " (torch layers)")


def test_sdf_grid_matches_reference(golden):
    g = golden("g10_geometry")
    r = golden("g9b_render")
    surf = build_surface(r)
    vols = [r[f"vol{i}"].cuda() for i in range(3)]
    u = surf.sdf_grid(vols, torch.tensor([-1.0, -1, -1]), torch.tensor([1.0, 1, 1]), int(g["resolution"]), chunk=100000)
    close(u, g["u"], atol=2e-5, rtol=1e-4, what="-sdf lattice")


@pytest.mark.parametrize("tag,chunk,precision", [("g15_validate", 512, "f32"), ("g15_validate", 8192, "f32"), ("g15_validate", 256, "f16x2"),
                                                 ("g15b_validate_l5", 512, "f32"), ("g15b_validate_l5", 8192, "f16x2")])
def test_validate_matches_the_reference_validate(golden, tag, chunk, precision):
    """The reference's own ImplicitSurface.validate (implicit_surface.py:429-470; three 256-ray chunks, golden g15) against the
    fused inference path with a different chunking: the jitter of every ray is the reference's (reference_jitter reproduces its
    chunk-by-chunk draws), colour / depth L1 within the north-star 1e-4, the images with their * 256 / * 128 + 128 scalings and
    clips (Q15), the SDF lattice handed to the iso-surface extraction.  g15b: the shipped five volume levels and four source views, i.e.
    the sdf_mlp_k<100> / blend S = 4 instantiations of the fused kernels."""
    g = golden(tag)
    gg = dict(g)
    gg["step"] = torch.tensor(-1.0)
    surf = build_surface(gg)
    feats, vols, masks, match, _ = scene_inputs(gg)
    c = lambda t: t.cuda()  # noqa: E731
    surf.val_chunk = chunk
    surf.sdf_precision = precision
    h, w = (int(x) for x in g["hw"])
    torch.manual_seed(int(g["rng_seed"]))
    out = surf.validate(c(g["rays_o"]), c(g["rays_d"]), c(g["near"]), c(g["far"]), vols, masks, c(g["imgs"]), feats, match, c(g["intrs"]), c(g["c2ws"]),
                        torch.tensor([-1.0, -1, -1]), torch.tensor([1.0, 1, 1]), torch.tensor([h, w]).int(), extract_geometry=True, mesh_resolution=33)
    assert set(out) == {"vertices", "triangles", "color_fine", "img_fine", "normal_img", "sdf_depth", "render_depth"}
    for k in ("color_fine", "render_depth", "sdf_depth"):
        assert tuple(out[k].shape) == tuple(g["out." + k].shape), k
        assert (torch.as_tensor(out[k]) - g["out." + k]).abs().mean() < 1e-4, k
    assert (torch.as_tensor(out["img_fine"]) - g["out.img_fine"]).abs().mean() < 256e-4
    assert (torch.as_tensor(out["normal_img"]) - g["out.normal_img"]).abs().mean() < 2e-2          # 128 x the normal's 1e-4
    assert out["img_fine"].min() >= 0 and out["img_fine"].max() <= 255 and out["normal_img"].min() >= 0 and out["normal_img"].max() <= 255
    close(torch.as_tensor(out["sdf_depth"]) > 0, g["out.sdf_depth"] > 0, atol=0, rtol=0, what="rays with a surface crossing", frac=0.005)
    # the lattice the reference hands to marching cubes, and a mesh through it
    u = surf.sdf_grid(vols, torch.tensor([-1.0, -1, -1]), torch.tensor([1.0, 1, 1]), 33)
    close(u, g["u"], atol=2e-5, rtol=1e-4, what="-sdf lattice")
    assert out["vertices"].shape[1] == 3 and out["triangles"].shape[1] == 3 and len(out["triangles"]) > 0
    assert out["vertices"].min() >= -1.0 - 1e-6 and out["vertices"].max() <= 1.0 + 1e-6


def test_partition_invariance_of_validate(golden):
    """Rays are independent: any chunking of validate() renders the same image (property tests)."""
    g = golden("g9a_render")
    surf = build_surface(g)
    feats, vols, masks, match, step = scene_inputs(g)
    c = lambda t: t.cuda()  # noqa: E731
    outs = []
    for chunk in (5, 24):
        surf.val_chunk = chunk
        torch.manual_seed(3)
        o = surf.validate(c(g["rays_o"]), c(g["rays_d"]), c(g["near"]), c(g["far"]), vols, masks, c(g["imgs"]), feats, match, c(g["intrs"]),
                          c(g["c2ws"]), None, None, (4, 6), extract_geometry=False)
        outs.append(o)
    for k in ["color_fine", "sdf_depth", "render_depth", "normal_img"]:
        # not bit-identical: rocBLAS picks GEMM tilings by batch size, and the resampling amplifies that round-off
        close(torch.as_tensor(outs[0][k]), torch.as_tensor(outs[1][k]), atol=1e-4, rtol=1e-4, what=k)


def test_validate_with_split_half_sdf_matches_float32(golden):
    g = golden("g9a_render")
    surf = build_surface(g)
    feats, vols, masks, match, step = scene_inputs(g)
    c = lambda t: t.cuda()  # noqa: E731
    outs = {}
    for prec in ("f32", "f16x2"):
        surf.sdf_precision = prec
        torch.manual_seed(3)
        outs[prec] = surf.validate(c(g["rays_o"]), c(g["rays_d"]), c(g["near"]), c(g["far"]), vols, masks, c(g["imgs"]), feats, match,
                                   c(g["intrs"]), c(g["c2ws"]), None, None, (4, 6), extract_geometry=False)
    for k in ["color_fine", "render_depth", "sdf_depth"]:
        a, b = torch.as_tensor(outs["f16x2"][k]), torch.as_tensor(outs["f32"][k])
        assert (a - b).abs().mean() < 1e-4, k


def test_split_half_overflow_renders_the_image_again_in_float32_with_the_same_jitter(golden):
    """A volume feature beyond the half range raises the split-half kernel's overflow flag; validate() then renders the SAME image (the
    jitter already drawn, not a fresh draw from the generator) with the float32 kernel."""
    g = golden("g9a_render")
    surf = build_surface(g)
    feats, vols, masks, match, step = scene_inputs(g)
    vols = [v.clone() for v in vols]
    vols[1][0, 2, 3:9, 3:9, 3:9] = 4.0e4                                          # > 3e4: not representable as a half pair
    c = lambda t: t.cuda()  # noqa: E731
    outs = {}
    flagged = []
    probe = surf._split_half_overflowed
    surf._split_half_overflowed = lambda: flagged.append(probe()) or flagged[-1]
    for prec in ("f32", "f16x2"):
        surf.sdf_precision = prec
        torch.manual_seed(3)
        outs[prec] = surf.validate(c(g["rays_o"]), c(g["rays_d"]), c(g["near"]), c(g["far"]), vols, masks, c(g["imgs"]), feats, match,
                                   c(g["intrs"]), c(g["c2ws"]), None, None, (4, 6), extract_geometry=False)
        after = torch.rand(1)                                                     # the generator is where the reference would leave it
        outs[prec + ".next_draw"] = after
    assert flagged == [False, True]                                               # the float32 run never asks; the split-half run overflowed
    assert torch.equal(outs["f32.next_draw"], outs["f16x2.next_draw"])
    for k in ["color_fine", "render_depth", "sdf_depth", "normal_img"]:
        close(torch.as_tensor(outs["f16x2"][k]), torch.as_tensor(outs["f32"][k]), atol=1e-6, rtol=1e-6, what=k)


def test_render_config0_coarsest_volume_only(golden):
    """BASELINE config[0] as written (golden g9d: 3 views 480 x 640, ONE 16^3 volume built from the level-4 map with intrinsics * 2^-4,
    512 rays): K1 on the device (mask bit-exact), then render() on the single-level pyramid against the reference's outputs -- through the FUSED
    kernels (round 5: gens_sdf_value / gens_sdf_grad / gens_sdf_train_* are built for one to five levels; until then a single level ran the
    PyTorch layers on the stand-alone K2 / K2'' kernels): the test asserts which entry points were launched, not only the numbers."""
    from gens_amd import ops
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.implicit_surface import ImplicitSurface
    from .test_oracle_golden import config0_scene
    g = golden("g9d_config0")
    sc = config0_scene(g)
    c = lambda t: t.cuda()  # noqa: E731
    intr4 = sc["intrs"].clone()
    intr4[:, :2] *= 0.5 ** 4
    _, masks = ops.volume_build([c(sc["features"][4])], c(intr4), c(sc["c2ws"]), [16])
    assert torch.equal(masks[0].cpu(), g["mask0"])
    surf = ImplicitSurface(gens_model_conf(volume_dims=(16,))["implicit_surface"])
    surf.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")}, strict=True)
    surf = surf.cuda()
    feats = [c(f) for f in sc["features"]]
    from gens_amd import lib as L
    torch.manual_seed(int(g["rng_seed"]))
    L.profile_begin()
    out = surf.render(c(g["rays_o"]), c(g["rays_d"]), c(sc["near"]), c(sc["far"]), [c(g["vol0"])], masks, c(sc["imgs"]), feats, feats,
                      c(sc["intrs"]), c(sc["c2ws"]), 1.0, None)
    launched = set(L.profile_end())
    assert {"gens_sdf_train_fwd", "gens_blend_train_fwd", "gens_composite_fwd"} <= launched, launched          # the fused training-mode path
    assert not ({"gens_lookup_volume_fwd", "gens_lookup_volume_bwd", "gens_lookup_volume_bwd2"} & launched), launched      # ... not the layers on K2 / K2''
    for k in ("color_fine", "render_depth", "sdf_depth"):
        assert (out[k].cpu() - g["out." + k]).abs().mean() < 1e-4, k
    assert (out["valid_mask"].cpu() != g["out.valid_mask"]).float().mean() < 0.005
    # the inference path of the same scene (validate's lean render): gens_sdf_value in the sampling rounds, gens_sdf_grad in render_core
    with torch.no_grad():
        torch.manual_seed(int(g["rng_seed"]))
        L.profile_begin()
        lean = surf.render(c(g["rays_o"]), c(g["rays_d"]), c(sc["near"]), c(sc["far"]), [c(g["vol0"])], masks, c(sc["imgs"]), feats, feats,
                           c(sc["intrs"]), c(sc["c2ws"]), 1.0, None, lean=True)
        launched = set(L.profile_end())
    assert {"gens_sdf_value", "gens_sdf_grad"} <= launched and any(k.startswith("gens_blend_views") for k in launched), launched
    assert "gens_lookup_volume_fwd" not in launched and "gens_lookup_feature_fwd" not in launched, launched
    for k in ("color_fine", "render_depth", "sdf_depth"):
        assert (lean[k].cpu() - g["out." + k]).abs().mean() < 1e-4, k
    out = surf.render_core(c(g["rays_o"]), c(g["rays_d"]), c(g["z_final"]), 2.0 / 64, [c(g["vol0"])], masks, feats, feats, c(sc["imgs"]),
                           c(sc["intrs"]), c(sc["c2ws"]), 1.0, None, pts_random=c(g["draw_ptsrand"]) * 2 - 1)
    assert torch.equal(out["valid_mask"].cpu(), g["out.valid_mask"])
    close(out["inside_sphere"], g["out.inside_sphere"], atol=0, rtol=0, what="inside_sphere")
    for k in ["weights", "weight_sum", "weight_max", "normal", "sdf_depth", "color_fine", "render_depth"]:
        close(out[k], g["out." + k], atol=1e-4, rtol=1e-3, what=k)
    close(out["gradients"], g["out.gradients"], atol=5e-4, rtol=1e-3, what="gradients")
    close(out["sparse_sdf"], g["out.sparse_sdf"], atol=1e-4, rtol=1e-4, what="sparse_sdf")
    for k in ["gradient_error", "smooth_error", "tv_reg"]:
        close(out[k], g["out." + k], atol=1e-4, rtol=2e-3, what=k)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ray_and_lattice_shards_reproduce_the_single_gpu_result_bit_for_bit(golden, world):
    """BASELINE config 4 on the HIP kernels: validate() with the rays split into `world` contiguous ranges and the lattice into chunks
    `index mod world` -- each shard rendered by the device kernels, gathered by the collective-free stand-in of the RCCL all_gather
    (gens_amd.distributed.Shard.single; the collectives themselves are covered on gloo, tests/test_distributed_cpu.py) -- must give
    the image and the lattice of the unsharded call EXACTLY: every ray's jitter is drawn on every rank from the same generator state and
    the fused kernels evaluate each point independently of its neighbours in the batch."""
    from gens_amd.distributed import Shard
    g = golden("g15_validate")
    gg = dict(g)
    gg["step"] = torch.tensor(-1.0)
    surf = build_surface(gg)
    feats, vols, masks, match, _ = scene_inputs(gg)
    c = lambda t: t.cuda()  # noqa: E731
    surf.val_chunk = 200                                     # not a divisor of any shard: chunk boundaries differ from the unsharded run
    h, w = (int(x) for x in g["hw"])
    bmin, bmax = torch.tensor([-1.0, -1, -1]), torch.tensor([1.0, 1, 1])
    args = (c(g["rays_o"]), c(g["rays_d"]), c(g["near"]), c(g["far"]), vols, masks, c(g["imgs"]), feats, match, c(g["intrs"]), c(g["c2ws"]), bmin, bmax,
            torch.tensor([h, w]).int())
    torch.manual_seed(77)
    ref = surf.validate(*args, extract_geometry=True, mesh_resolution=33)
    ref_u = surf.sdf_grid(vols, bmin, bmax, 33, chunk=5000)
    sink, out = {}, None
    for r in range(world):
        torch.manual_seed(77)                                # every rank starts from the same CPU generator state
        out = surf.validate(*args, extract_geometry=True, mesh_resolution=33, shard=Shard.single(r, world, sink))
        assert (out is None) == (r + 1 < world)
    for k in ("color_fine", "img_fine", "normal_img", "sdf_depth", "render_depth", "vertices", "triangles"):
        assert torch.equal(torch.as_tensor(out[k]), torch.as_tensor(ref[k])), k
    sink, u = {}, None
    for r in range(world):
        u = surf.sdf_grid(vols, bmin, bmax, 33, chunk=5000, shard=Shard.single(r, world, sink))
    assert torch.equal(u, ref_u)


def test_split_half_overflow_recomputes_the_lattice_in_float32(golden):
    """An out-of-range volume feature raises the split-half kernel's overflow flag during the SDF LATTICE too: extract_geometry must then
    hand marching cubes float32 values (the flag used to be consumed by the image check only, leaving a mesh from out-of-range halves)."""
    g = golden("g9a_render")
    surf = build_surface(g)
    feats, vols, masks, match, step = scene_inputs(g)
    vols = [v.clone() for v in vols]
    vols[1][0, 2, 3:9, 3:9, 3:9] = 4.0e4
    bmin, bmax = torch.tensor([-1.0, -1, -1]), torch.tensor([1.0, 1, 1])
    surf.sdf_precision = "f32"
    ref = surf.sdf_grid(vols, bmin, bmax, 24)
    surf.sdf_precision = "f16x2"
    got = surf.sdf_grid(vols, bmin, bmax, 24)
    assert torch.equal(got, ref)
    assert not surf._split_half_overflowed()                  # the lattice consumed its own flag: nothing left for the image check
    c = lambda t: t.cuda()  # noqa: E731
    torch.manual_seed(3)
    out = surf.validate(c(g["rays_o"]), c(g["rays_d"]), c(g["near"]), c(g["far"]), vols, masks, c(g["imgs"]), feats, match, c(g["intrs"]), c(g["c2ws"]),
                        bmin, bmax, (4, 6), extract_geometry=True, mesh_resolution=24)
    surf.sdf_precision = "f32"
    torch.manual_seed(3)
    ref_out = surf.validate(c(g["rays_o"]), c(g["rays_d"]), c(g["near"]), c(g["far"]), vols, masks, c(g["imgs"]), feats, match, c(g["intrs"]),
                            c(g["c2ws"]), bmin, bmax, (4, 6), extract_geometry=True, mesh_resolution=24)
    assert torch.equal(torch.as_tensor(out["vertices"]), torch.as_tensor(ref_out["vertices"]))
    close(torch.as_tensor(out["color_fine"]), torch.as_tensor(ref_out["color_fine"]), atol=1e-6, rtol=1e-6, what="colour")


def test_head_start_on_the_next_images_jitter_changes_nothing(golden):
    """validate() starts the NEXT image's jitter draws itself (on a private generator, used only if the default generator has not moved): images
    rendered one after the other equal the ones rendered without the head start, the generator ends in the same state, and a draw between two
    images (a training step's) makes the head start void -- the stream everybody sees is the reference's.  prefetch_jitter() (rounds 2 - 5's
    explicit call) is the same mechanism."""
    g = golden("g9a_render")
    surf = build_surface(g)
    assert surf.val_chunk is None and surf.speculate_jitter          # the shipped defaults are what is tested
    feats, vols, masks, match, step = scene_inputs(g)
    c = lambda t: t.cuda()  # noqa: E731
    args = (c(g["rays_o"]), c(g["rays_d"]), c(g["near"]), c(g["far"]), vols, masks, c(g["imgs"]), feats, match, c(g["intrs"]), c(g["c2ws"]), None, None, (4, 6))

    def sequence(explicit=False):
        torch.manual_seed(11)
        imgs = [surf.validate(*args, extract_geometry=False)]
        if explicit:
            surf.prefetch_jitter(24)
        imgs.append(surf.validate(*args, extract_geometry=False))
        between = torch.rand(5)                                       # somebody else draws (a training step's t_rand): the head start is void
        imgs.append(surf.validate(*args, extract_geometry=False))
        return imgs, between, torch.rand(1)

    surf.speculate_jitter = False
    a, a_between, end_a = sequence()
    assert getattr(surf, "_jitter_ahead", None) is None
    surf.speculate_jitter = True
    b, b_between, end_b = sequence()
    assert surf._jitter_ahead is not None                            # (a head start for a fourth image is pending: dropped below)
    e, e_between, end_e = sequence(explicit=True)
    surf.join_speculation()
    for other, bt, end in ((b, b_between, end_b), (e, e_between, end_e)):
        for k in ("color_fine", "sdf_depth", "render_depth"):
            for x, y in zip(a, other):
                assert torch.equal(torch.as_tensor(x[k]), torch.as_tensor(y[k])), k
        assert torch.equal(a_between, bt) and torch.equal(end_a, end)
    assert not torch.equal(torch.as_tensor(a[0]["color_fine"]), torch.as_tensor(a[1]["color_fine"]))      # every image has its own jitter
