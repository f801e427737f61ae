"""GPU property tests at BASELINE.json's FULL sizes (5 views, 480x640, volume_dims 256/128/64[/32/16], 128 samples per ray),
where the CPU oracle is too slow to be the checker: size-independent properties of the domain instead."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    from gens_amd import ops, synthetic
    sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
    dev = torch.device("cuda")
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
    d["features"] = [f.to(dev) for f in sc["features"]]
    d["dims"] = [256, 128, 64, 32, 16]
    with torch.no_grad():
        d["cost"], d["masks"] = ops.volume_build(d["features"], d["intrs"], d["c2ws"], d["dims"])
    d["vols"] = [v.to(dev) for v in synthetic.make_volumes(d["dims"], seed=1)]
    ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640)
    d["rays_o"], d["rays_d"] = ro.to(dev), rd.to(dev)
    return d


def test_k1_full_size_statistics_and_view_permutation(scene):
    """mean/variance over views do not depend on the order of the views; var >= -eps; empty voxels are exactly zero;
    mask == (count >= 2) is monotone: removing a view can only clear mask voxels."""
    from gens_amd import ops
    cost, masks = scene["cost"], scene["masks"]
    perm = [0, 3, 1, 4, 2]
    feats_p = [f[perm].contiguous() for f in scene["features"]]
    cost_p, masks_p = ops.volume_build(feats_p, scene["intrs"][perm].contiguous(), scene["c2ws"][perm].contiguous(), scene["dims"])
    for lvl in range(5):
        assert torch.equal(masks[lvl], masks_p[lvl])
        assert (cost[lvl] - cost_p[lvl]).abs().max() < 2e-5          # summation order only
        assert cost[lvl][:, 4:].min() > -1e-4                        # E[x^2]-E[x]^2 in float32
        assert masks[lvl].min() >= 0 and masks[lvl].max() <= 1
    frac = float(masks[0].mean())
    assert 0.2 < frac < 0.8, frac
    _, masks4 = ops.volume_build([f[:4].contiguous() for f in scene["features"]], scene["intrs"][:4].contiguous(),
                                 scene["c2ws"][:4].contiguous(), scene["dims"])
    for lvl in range(5):
        assert (masks4[lvl] <= masks[lvl]).all()
    empty = (cost[0][:, :4].abs().sum(1, keepdim=True) == 0) & (masks[0] == 0)
    assert (cost[0][:, 4:][empty.expand(-1, 4, -1, -1, -1)] == 0).all()


def test_k2_full_size_linearity_and_lattice_identity(scene):
    """The look-up is linear in the volume, exact at voxel centres, and packed == planar layout."""
    from gens_amd import ops
    vols = scene["vols"]
    g = torch.Generator(device="cuda").manual_seed(5)
    pts = torch.rand(2_000_000, 3, device="cuda", generator=g) * 2 - 1
    a = ops.lookup_volume(pts, vols)
    b = ops.lookup_volume(pts, ops.VolumeSet.packed(vols))
    assert torch.equal(a, b)
    other = [torch.randn_like(v) for v in vols]
    lin = ops.lookup_volume(pts, [2.0 * v + 0.5 * o for v, o in zip(vols, other)])
    assert (lin - (2.0 * a + 0.5 * ops.lookup_volume(pts, other))).abs().max() < 1e-4
    d = 256
    idx = torch.randint(0, d, (100000, 3), device="cuda", generator=g)
    centres = idx.float() / (d - 1) * 2 - 1
    at = ops.lookup_volume(centres, [vols[0]])
    ref = vols[0][0][:, idx[:, 0], idx[:, 1], idx[:, 2]].t()
    assert (at - ref).abs().max() < 2e-5


def test_sampling_full_image_sorted_bounded_and_new_samples_inside_their_bins(scene):
    from gens_amd import ops
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.implicit_surface import ImplicitSurface, Scene
    torch.manual_seed(0)
    surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(scene["dims"]))["implicit_surface"]).cuda().eval()
    sc = Scene(scene["vols"], scene["masks"], scene["imgs"], scene["features"], scene["features"], scene["intrs"], scene["c2ws"])
    n = 65536
    ro, rd = scene["rays_o"][100000:100000 + n].contiguous(), scene["rays_d"][100000:100000 + n].contiguous()
    z0 = (scene["near"] + (scene["far"] - scene["near"]) * torch.linspace(0, 1, 64, device="cuda")[None]).expand(n, 64).contiguous()
    z = surf._sample_rays(ro, rd, z0, sc)
    assert z.shape == (n, 128)
    assert (z[:, 1:] >= z[:, :-1]).all()                                     # sorted
    assert z.min() >= z0.min() - 1e-6 and z.max() <= z0.max() + 1e-6          # importance samples stay inside [near, far]
    # every coarse sample survives the four merges
    pos = torch.searchsorted(z.contiguous(), z0.contiguous())
    assert torch.equal(torch.gather(z, 1, pos.clamp(max=127)), z0)


def test_fused_sampling_rounds_equal_the_separate_operators_bit_for_bit(scene):
    """One launch per sampling round (gens_merge_upsample: cat_z_vals of round i + up_sample / sample_pdf of round i + 1, and after the last
    round the merge + render_core's section mid-points) against the separate operators (gens_merge_samples, gens_upsample, gens_ray_points):
    the same samples, points and mask decisions, bit for bit, on a full ray chunk with jitter (ties and rays without any valid sample included)."""
    from gens_amd import ops
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.implicit_surface import ImplicitSurface, Scene
    torch.manual_seed(0)
    dims = scene["dims"][:3]
    surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"]).cuda().eval()
    sc = Scene(scene["vols"][:3], scene["masks"][:3], scene["imgs"], scene["features"], scene["features"], scene["intrs"], scene["c2ws"])
    n = 32768 + 3                                                            # (a last block of three rays)
    ro, rd = scene["rays_o"][90000:90000 + n].contiguous(), scene["rays_d"][90000:90000 + n].contiguous()
    z0 = (scene["near"] + (scene["far"] - scene["near"]) * torch.linspace(0, 1, 64, device="cuda")[None]).expand(n, 64)
    z0 = (z0 + (torch.rand(n, 1, generator=torch.Generator().manual_seed(3)).cuda() - 0.5) * 2.0 / 64).contiguous()
    with torch.no_grad():
        surf.fused_sampling = False
        z_sep = surf._sample_rays(ro, rd, z0, sc)
        pts_sep, valid_sep = ops.ray_points(ro, rd, z_sep, sc.masks, mid=True, sample_dist=2.0 / 64)
        surf.fused_sampling = True
        z_fused = surf._sample_rays(ro, rd, z0, sc)
        assert surf._mid_points is None
        z_fused2 = surf._sample_rays(ro, rd, z0, sc, mid_points=2.0 / 64)
        cached = surf._mid_points
    assert torch.equal(z_sep, z_fused) and torch.equal(z_sep, z_fused2)
    assert cached is not None and cached[0] is z_fused2 and torch.equal(cached[2], pts_sep) and torch.equal(cached[3], valid_sep)
    # a single round through the operators themselves, with mask decisions and duplicate z values (ties keep the older sample first)
    g = torch.Generator().manual_seed(5)
    b, m = 1000, 80
    z = torch.sort(torch.rand(b, m, generator=g) * 2 + 1, dim=1)[0].cuda()
    z_add = torch.sort(torch.rand(b, 16, generator=g) * 2 + 1, dim=1)[0].cuda()
    z_add[:, 3] = z[:, 40]                                                   # ties between an old and a new sample, and between two new ones
    z_add[:, 4] = z_add[:, 3]
    z_add = torch.sort(z_add, dim=1)[0].contiguous()
    sdf, sdf_add = torch.randn(b, m, generator=g).cuda() * 0.1, torch.randn(b, 16, generator=g).cuda() * 0.1
    valid, valid_add = (torch.rand(b, m, generator=g) < 0.7).cuda(), (torch.rand(b, 16, generator=g) < 0.7).cuda()
    valid[:7] = False
    valid_add[:7] = False
    rays_o, rays_d = ro[:b].contiguous(), rd[:b].contiguous()
    zm, sm, vm = ops.merge_samples(z, z_add, sdf, sdf_add, valid, valid_add)
    zn, pn, vn = ops.upsample(rays_o, rays_d, zm, sm, 16, sc.masks, 256.0, valid_in=vm)
    fz, fs, fv, fzn, fpn, fvn = ops.merge_upsample(rays_o, rays_d, z, sdf, valid, z_add, sdf_add, valid_add, 16, sc.masks, 256.0)
    for a, c in ((zm, fz), (sm, fs), (vm, fv), (zn, fzn), (pn, fpn), (vn, fvn)):
        assert torch.equal(a, c)


def test_render_full_chunk_weights_are_a_sub_probability_and_partition_invariant(scene):
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.implicit_surface import ImplicitSurface, Scene
    torch.manual_seed(0)
    dims = scene["dims"][:3]
    surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"]).cuda().eval()
    sc = Scene(scene["vols"][:3], scene["masks"][:3], scene["imgs"], scene["features"], scene["features"], scene["intrs"], scene["c2ws"])
    n = 32768
    ro, rd = scene["rays_o"][120000:120000 + n].contiguous(), scene["rays_d"][120000:120000 + n].contiguous()
    t_rand = torch.rand(n, 1, generator=torch.Generator().manual_seed(1))
    args = (scene["near"], scene["far"], scene["vols"][:3], scene["masks"][:3], scene["imgs"], scene["features"], scene["features"],
            scene["intrs"], scene["c2ws"], 1.0, None)
    with torch.no_grad():
        full = surf.render(ro, rd, *args, scene=sc, lean=True, t_rand=t_rand)
        parts = [surf.render(ro[s:s + 8192], rd[s:s + 8192], *args, scene=sc, lean=True, t_rand=t_rand[s:s + 8192]) for s in range(0, n, 8192)]
    w = full["weights"]
    assert w.min() >= 0 and w.max() <= 1 + 1e-6
    assert full["weight_sum"].max() <= 1 + 1e-4                              # transmittance never goes negative
    assert torch.isfinite(full["color_fine"]).all() and full["color_fine"].min() >= -1e-5 and full["color_fine"].max() <= 1 + 1e-4
    inside = full["inside_sphere"]
    assert ((inside == 0) | (inside == 1)).all()
    for k in ("color_fine", "render_depth", "sdf_depth", "weights"):
        cat = torch.cat([p[k] for p in parts], 0)
        assert (cat - full[k]).abs().max() < 1e-5, k                         # rays are independent (fused kernels: batch-shape free)


def test_blend_is_invariant_to_source_view_order(scene):
    """Softmax / weighted mean / variance over source views cannot depend on the order of the sources."""
    from gens_amd import ops
    from gens_amd.models.modules.blending_network import BlendingNetwork
    torch.manual_seed(2)
    net = BlendingNetwork(d_feature=20).cuda()
    g = torch.Generator(device="cuda").manual_seed(3)
    pts = torch.rand(200000, 3, device="cuda", generator=g) * 1.6 - 0.8
    views = ops.SceneViews(scene["imgs"], scene["intrs"], scene["c2ws"], scene["features"])
    perm = [0, 4, 2, 1, 3]
    views_p = ops.SceneViews(scene["imgs"][perm].contiguous(), scene["intrs"][perm].contiguous(), scene["c2ws"][perm].contiguous(),
                             [f[perm].contiguous() for f in scene["features"]])
    plan = ops.BlendPlan(net)
    rgb, vis = ops.blend_views(plan, views, pts)
    rgb_p, vis_p = ops.blend_views(plan, views_p, pts)
    assert (rgb - rgb_p).abs().max() < 2e-5
    assert torch.equal(vis[:, [3, 1, 0, 2]], vis_p)                          # source k of the permuted scene is view perm[k+1]


def test_k1_exact_division_shortcuts_cover_every_float32():
    """RN(1/b) = v_rcp_f32 + one FMA refinement, and a/b = reciprocal multiply + FMA correction, against the IEEE division for all
    2^32 bit patterns of b (and one hashed numerator each): not one mismatch, so K1's quotients are the reference's."""
    from gens_amd import lib as L
    counts = torch.zeros(2, dtype=torch.int64, device="cuda")
    L.call("gens_selftest_division", L.ptr(counts, torch.int64), L.stream())
    assert counts.tolist() == [0, 0]


def test_k1_fast_path_is_bit_identical_to_the_generic_kernel(scene):
    """The power-of-two forward kernel replaces IEEE divisions by reciprocal + FMA-corrected quotients and index divisions by
    shifts; every output must equal the generic kernel's bit for bit (full size: 19 M voxels x 5 views, incl. border voxels)."""
    import os
    from gens_amd import ops
    feats, intrs, c2ws = scene["features"], scene["intrs"], scene["c2ws"]
    dims = [256, 128, 64, 32, 16]
    fast_v, fast_m = ops.volume_build(feats, intrs, c2ws, dims)          # production: one launch, culled, exact-division shortcuts
    outs = []
    for switch in ("GENS_K1_SINGLE", "GENS_K1_GENERIC"):                 # the previous power-of-two kernel, then the plain IEEE kernel
        os.environ[switch] = "1"
        try:
            outs.append(ops.volume_build(feats, intrs, c2ws, dims))
        finally:
            del os.environ[switch]
    for ref_v, ref_m in outs:
        for a, b in zip(fast_v + fast_m, ref_v + ref_m):
            assert torch.equal(a, b)
    # a non-pinhole camera (skewed intrinsics, projective last row of w2c is still 0 0 0 1) takes the general branch per view
    skew = intrs.clone()
    skew[1:, 0, 1] = 3.0
    skew[2, 1, 3] = 0.25
    fast = ops.volume_build(feats, skew, c2ws, dims[:3])
    os.environ["GENS_K1_GENERIC"] = "1"
    try:
        ref = ops.volume_build(feats, skew, c2ws, dims[:3])
    finally:
        del os.environ["GENS_K1_GENERIC"]
    for a, b in zip(fast[0] + fast[1], ref[0] + ref[1]):
        assert torch.equal(a, b)
    # the production kernel culls (z-row, view) pairs from a conservative frustum interval: cameras inside / beside / behind the cube,
    # rolled and tilted, narrow and wide fields of view -- every voxel the exact test accepts must survive the culling
    g = torch.Generator().manual_seed(11)
    seen = 0.0
    for trial in range(6):
        rig_c2w, rig_k = [], []
        for v in range(5):
            q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
            if torch.det(q) < 0:
                q[:, 0] = -q[:, 0]
            c2w = torch.eye(4, dtype=torch.float64)
            c2w[:3, :3] = q
            c2w[:3, 3] = torch.randn(3, generator=g, dtype=torch.float64) * [0.3, 1.0, 2.5][trial % 3]
            k = intrs[0].double().clone()
            k[0, 0] *= [0.2, 1.0, 3.0][(trial + v) % 3]
            k[1, 1] *= [0.2, 1.0, 3.0][(trial + v) % 3]
            rig_c2w.append(c2w)
            rig_k.append(k)
        rig_c2w, rig_k = torch.stack(rig_c2w).float().cuda(), torch.stack(rig_k).float().cuda()
        fast = ops.volume_build(feats, rig_k, rig_c2w, dims[:4])
        os.environ["GENS_K1_GENERIC"] = "1"
        try:
            ref = ops.volume_build(feats, rig_k, rig_c2w, dims[:4])
        finally:
            del os.environ["GENS_K1_GENERIC"]
        seen += sum(float(m.sum()) for m in ref[1])
        for a, b in zip(fast[0] + fast[1], ref[0] + ref[1]):
            assert torch.equal(a, b)
    assert seen > 1e5                                                        # the rigs do see the cube
    # structured worst cases for the interval arithmetic: optical axis along the z-rows (image coordinates constant along a row),
    # perpendicular to them, a camera inside the cube, one in a corner looking at the far corner, one looking away
    def look(eye, target, up=(0.0, 1.0, 0.0)):
        eye, target, up = (torch.tensor(v, dtype=torch.float64) for v in (eye, target, up))
        z = (target - eye) / torch.linalg.norm(target - eye)
        x = torch.linalg.cross(up, z)
        x = x / torch.linalg.norm(x)
        y = torch.linalg.cross(z, x)
        c2w = torch.eye(4, dtype=torch.float64)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, eye
        return c2w
    rig = torch.stack([look((0.0, 0.0, -2.2), (0.0, 0.0, 0.0)), look((2.2, 0.0, 0.0), (0.0, 0.0, 0.0)), look((0.1, -0.2, 0.3), (1.0, 0.3, -0.5)),
                       look((1.0, 1.0, 1.0), (-1.0, -1.0, -1.0), up=(0.0, 0.0, 1.0)), look((0.0, 0.0, -2.2), (0.0, 0.0, -5.0))]).float().cuda()
    fast = ops.volume_build(feats, intrs, rig, dims[:4])
    os.environ["GENS_K1_GENERIC"] = "1"
    try:
        ref = ops.volume_build(feats, intrs, rig, dims[:4])
    finally:
        del os.environ["GENS_K1_GENERIC"]
    assert float(ref[1][0].mean()) > 0.05
    for a, b in zip(fast[0] + fast[1], ref[0] + ref[1]):
        assert torch.equal(a, b)


def _k1_bwd(L, tex, w2c, intrs, scale, d, gvol, how):
    """gens_volume_build_bwd through the C ABI: how = "window" (the wave-window kernel) or "direct" (every tap a global atomic)."""
    import os
    nv, h, w, _ = tex.shape
    gf = torch.zeros_like(tex)
    if how == "direct":
        os.environ["GENS_K1_BWD_DIRECT"] = "1"
    try:
        L.call("gens_volume_build_bwd", L.ptr(tex), L.ptr(w2c), L.ptr(intrs), scale, nv, h, w, d, L.ptr(gvol), L.ptr(gf), L.stream())
    finally:
        os.environ.pop("GENS_K1_BWD_DIRECT", None)
    return gf


@pytest.mark.parametrize("dims", [[128, 64, 32], [32, 16, 8], [24, 16]])
def test_k1_leaves_the_masks_as_bits_too(dims):
    """gens_volume_build_levels_bits: the words the volume build writes are gens_pack_mask_bits of its float masks (tiled levels, small levels, the
    level-by-level fallback of a side that is not a power of two; tiles no view reaches included), and VolumeSet.bit_table takes them from the mask
    tensors instead of packing again."""
    from gens_amd import lib as L, ops, synthetic
    from gens_amd.ops.base import VolumeSet
    sc = synthetic.make_scene(nv=3, h=96, w=128, n_levels=len(dims), seed=5)
    feats = [f.cuda() for f in sc["features"][:len(dims)]]
    with torch.no_grad():
        vols, masks = ops.volume_build(feats, sc["intrs"].cuda(), sc["c2ws"].cuda(), dims)
    for m, d in zip(masks, dims):
        n = d ** 3
        want = torch.empty((n + 31) // 32, device="cuda", dtype=torch.int32)
        L.call("gens_pack_mask_bits", L.ptr(m.reshape(-1).contiguous()), n, L.ptr(want, torch.int32), L.stream())
        ver, got = m._gens_bits
        assert ver == m._version and torch.equal(got, want), d
        assert 0 < int((m > 0).sum()) < n                       # (both kinds of tiles are in the test)
    vs = VolumeSet.masks(masks)
    L.profile_begin()
    vs.bit_table()
    assert not any(name == "gens_pack_mask_bits" for name, _, _, _ in L.profile_end(raw=True))


def test_k1_backward_window_kernel_equals_direct_atomics(scene):
    """The LDS-window scatter (one global atomic per touched texel and channel of a wave's 4 x 16 voxel tile) against one global atomic per tap, at full
    size (19 M voxels x 5 views): the same sums in a different order."""
    from gens_amd import lib as L, ops
    feats, intrs, c2ws = scene["features"], scene["intrs"], scene["c2ws"]
    w2c = torch.linalg.inv(c2ws).contiguous()
    g = torch.Generator(device="cuda").manual_seed(5)
    for lvl, d in [(0, 256), (1, 128), (2, 64), (3, 32), (0, 32), (0, 16), (1, 48), (2, 20)]:
        tex = ops.pack_nchw(feats[lvl])
        gvol = torch.randn(8, d, d, d, device="cuda", generator=g)
        ref = _k1_bwd(L, tex, w2c, intrs, 0.5 ** lvl, d, gvol, "direct")
        scale = float(ref.abs().max())
        assert scale > 1.0
        out = _k1_bwd(L, tex, w2c, intrs, 0.5 ** lvl, d, gvol, "window")
        assert float((out - ref).abs().max()) <= 2e-6 * scale + 1e-6, (d, float((out - ref).abs().max()), scale)


def _k1_bwd_levels(L, texs, w2c, intrs, dims, gvols, min_vis=1):
    """gens_volume_build_levels (with the count planes) followed by gens_volume_build_bwd_levels through the C ABI; intrs: per level, rows 0-1 pre-scaled.
    gvols[l] = None: no gradient for that level."""
    n, nv = len(dims), texs[0].shape[0]
    hw = [x for t in texs for x in (t.shape[1], t.shape[2])]
    vols = [torch.empty(8, d, d, d, device="cuda") for d in dims]
    masks = [torch.empty(d, d, d, device="cuda") for d in dims]
    counts = [torch.empty(d ** 3, device="cuda", dtype=torch.uint8) for d in dims]
    L.call("gens_volume_build_levels", L.ptr_table(texs), L.int_table(hw), L.int_table(dims), n, L.ptr(w2c), L.ptr_table(intrs), nv, min_vis, L.ptr_table(vols),
           L.ptr_table(masks), L.ptr_table(counts, torch.uint8), L.stream())
    need = L.load().gens_volume_build_bwd_levels_scratch_bytes(L.int_table(hw), L.int_table(dims), n, nv)
    assert need > 0
    scratch = torch.empty(need, device="cuda", dtype=torch.uint8).fill_(0xA5)            # (contents irrelevant: the call initialises what it reads)
    out = [torch.zeros_like(t) for t in texs]
    L.call("gens_volume_build_bwd_levels", L.ptr_table(texs), L.int_table(hw), L.int_table(dims), n, L.ptr(w2c), L.ptr_table(intrs), nv, L.ptr_table(vols),
           L.ptr_table(counts, torch.uint8), L.ptr_table(gvols), L.ptr_table(out), L.ptr(scratch, torch.uint8), need, L.stream())
    return out, counts


def _scaled_intrinsics(intrs, n):
    out = []
    for l in range(n):
        k = intrs.clone()
        k[:, :2] *= 0.5 ** l
        out.append(k.contiguous())
    return out


def test_k1_backward_all_levels_in_one_launch_set_equals_direct_atomics(scene):
    """gens_volume_build_bwd_levels (means and visible-view counts from the forward pass, tile ranges from the corners of the 4 x 16 wave tiles, 64-bit
    double sums in LDS) against one global atomic per tap: the shipped five-level pyramid at full size in ONE call, a
    call with levels switched off, and single levels down to 16^3 (where most wave tiles span more than 2 x 2 image tiles: the direct bin)."""
    from gens_amd import lib as L, ops
    feats, intrs, c2ws = scene["features"], scene["intrs"], scene["c2ws"]
    w2c = torch.linalg.inv(c2ws).contiguous()
    g = torch.Generator(device="cuda").manual_seed(5)
    dims = [256, 128, 64, 32, 16]
    texs = [ops.pack_nchw(feats[l]) for l in range(5)]
    ks = _scaled_intrinsics(intrs, 5)
    gvols = [torch.randn(8, d, d, d, device="cuda", generator=g) for d in dims]
    refs = [_k1_bwd(L, texs[l], w2c, ks[l], 1.0, dims[l], gvols[l], "direct") for l in range(5)]
    out, counts = _k1_bwd_levels(L, texs, w2c, ks, dims, gvols)
    assert 0.5 < float(counts[0].float().mean()) < 4.5                                    # (the scene's volume is neither invisible nor seen by every view)
    for l in range(5):
        scale = float(refs[l].abs().max())
        assert scale > 1.0 and float((out[l] - refs[l]).abs().max()) <= 2e-6 * scale + 1e-6, (dims[l], float((out[l] - refs[l]).abs().max()), scale)
    part, _ = _k1_bwd_levels(L, texs, w2c, ks, dims, [None, gvols[1], None, gvols[3], None])
    for l in range(5):
        if l in (1, 3):
            assert float((part[l] - refs[l]).abs().max()) <= 2e-6 * float(refs[l].abs().max()) + 1e-6
        else:
            assert float(part[l].abs().max()) == 0.0
    for lvl, d in [(0, 32), (0, 16), (1, 48), (2, 16)]:                                   # other pairings of map and volume size
        gv = torch.randn(8, d, d, d, device="cuda", generator=g)
        ref = _k1_bwd(L, texs[lvl], w2c, ks[lvl], 1.0, d, gv, "direct")
        one, _ = _k1_bwd_levels(L, [texs[lvl]], w2c, [ks[lvl]], [d], [gv])
        assert float((one[0] - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-6, (lvl, d)


@pytest.mark.parametrize("nv,h,w,d", [(11, 50, 70, 48), (2, 31, 65, 32), (16, 24, 20, 16), (3, 200, 130, 64)])
def test_k1_backward_levels_other_view_counts_and_image_sizes(nv, h, w, d):
    """The all-level backward with up to GENS_MAX_VIEWS views (the forward pass then takes its generic kernel, which writes the counts too) and images
    that are not whole tiles (64 x 30): one partial tile, a tile row of one texel, an image smaller than a tile."""
    from gens_amd import lib as L, ops, synthetic
    sc = synthetic.make_scene(nv=nv, h=h, w=w, n_levels=1, seed=nv + d)
    tex = ops.pack_nchw(sc["features"][0].cuda())
    w2c = torch.linalg.inv(sc["c2ws"].cuda()).contiguous()
    intrs = sc["intrs"].cuda()
    g = torch.Generator(device="cuda").manual_seed(d)
    gvol = torch.randn(8, d, d, d, device="cuda", generator=g)
    ref = _k1_bwd(L, tex, w2c, intrs, 1.0, d, gvol, "direct")
    out, _ = _k1_bwd_levels(L, [tex], w2c, [intrs], [d], [gvol])
    scale = float(ref.abs().max())
    # (a 48^3 volume over a 50 x 70 image sums ~350 taps per texel and view: the float32 atomics of the direct scatter carry 4e-6 of the largest entry
    # in their order of arrival; the window's double sums are exact up to the final conversion)
    assert scale > 0.5 and float((out[0] - ref).abs().max()) <= 1e-5 * scale + 1e-6, (float((out[0] - ref).abs().max()), scale)


def test_k1_backward_levels_cameras_inside_and_behind_the_volume():
    """Wave tiles with a corner at or behind a camera (their image is not the corners' convex hull: the pair's 64 voxels are walked one by one) and
    views that see none of the volume: cameras moved into the cube and turned away from it."""
    from gens_amd import lib as L, ops, synthetic
    sc = synthetic.make_scene(nv=4, h=120, w=160, n_levels=1, seed=3)
    c2ws = sc["c2ws"].clone()
    c2ws[1, :3, 3] = torch.tensor([0.1, -0.2, 0.05])                                      # inside the cube
    c2ws[2, :3, 3] *= 0.45                                                                  # close to / inside a face
    c2ws[3, :3, :3] = c2ws[3, :3, :3] @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))        # looks away
    tex = ops.pack_nchw(sc["features"][0].cuda())
    w2c = torch.linalg.inv(c2ws.cuda()).contiguous()
    intrs = sc["intrs"].cuda()
    g = torch.Generator(device="cuda").manual_seed(2)
    for d in (32, 64):
        gvol = torch.randn(8, d, d, d, device="cuda", generator=g)
        ref = _k1_bwd(L, tex, w2c, intrs, 1.0, d, gvol, "direct")
        out, counts = _k1_bwd_levels(L, [tex], w2c, [intrs], [d], [gvol])
        scale = float(ref.abs().max())
        assert scale > 0.5 and float((out[0] - ref).abs().max()) <= 1e-5 * scale + 1e-6, (d, float((out[0] - ref).abs().max()), scale)
        assert float(ref[1].abs().max()) > 0 and float(out[0][3].abs().max()) == float(ref[3].abs().max())


def test_k1_backward_levels_sparse_and_nonfinite_gradients(scene):
    """A cotangent that is zero almost everywhere with entries 1e-30 .. 1e+30 (double sums: no scale to choose), an all-zero cotangent, and a NaN / an
    infinity among the cotangents (the same texels turn NaN / inf as with the wave-window kernel)."""
    from gens_amd import lib as L, ops
    feats, intrs, c2ws = scene["features"], scene["intrs"], scene["c2ws"]
    w2c = torch.linalg.inv(c2ws).contiguous()
    tex = ops.pack_nchw(feats[1])
    k1 = _scaled_intrinsics(intrs, 2)[1]
    d = 64
    g = torch.Generator(device="cuda").manual_seed(9)
    for amp in (1e-30, 1.0, 1e30):
        gvol = torch.zeros(8, d, d, d, device="cuda")
        gvol[:4, 20:24, 30:34, 8:40] = amp * torch.randn(4, 4, 4, 32, device="cuda", generator=g)       # (mean planes: finite at 1e30; the variance planes multiply by features)
        gvol[4:, 40, 20, 10:20] = amp * 1e-3
        ref = _k1_bwd(L, tex, w2c, k1, 1.0, d, gvol, "direct")
        out = _k1_bwd_levels(L, [tex], w2c, [k1], [d], [gvol])[0][0]
        scale = float(ref.abs().max())
        assert scale > 0.1 * amp and bool(torch.isfinite(ref).all())
        assert float((out - ref).abs().max()) <= 2e-6 * scale, (amp, float((out - ref).abs().max()), scale)
        assert torch.equal(out == 0, ref == 0)                                       # untouched texels stay exactly zero
    zero = _k1_bwd_levels(L, [tex], w2c, [k1], [d], [torch.zeros(8, d, d, d, device="cuda")])[0][0]
    assert float(zero.abs().max()) == 0.0
    for bad in (float("nan"), float("inf")):
        gvol = torch.randn(8, d, d, d, device="cuda", generator=g)
        gvol[2, 31, 33, 17] = bad
        ref = _k1_bwd(L, tex, w2c, k1, 1.0, d, gvol, "window")
        out = _k1_bwd_levels(L, [tex], w2c, [k1], [d], [gvol])[0][0]
        assert int((~torch.isfinite(ref)).sum()) > 0
        assert torch.equal(torch.isnan(out), torch.isnan(ref)) and torch.equal(torch.isinf(out), torch.isinf(ref))
        ok = torch.isfinite(ref)
        assert float((out[ok] - ref[ok]).abs().max()) <= 2e-6 * float(ref[ok].abs().max()) + 1e-6


def test_k1_backward_levels_rejects_what_it_does_not_cover(scene):
    from gens_amd import lib as L, ops
    lib = L.load()
    assert lib.gens_volume_build_bwd_levels_scratch_bytes(L.int_table([480, 640]), L.int_table([24]), 1, 5) == 0        # D must be a multiple of 16
    assert lib.gens_volume_build_bwd_levels_scratch_bytes(L.int_table([480, 640]), L.int_table([64]), 1, 17) == 0       # GENS_MAX_VIEWS
    need = lib.gens_volume_build_bwd_levels_scratch_bytes(L.int_table([240, 320]), L.int_table([64]), 1, 5)
    assert 0 < need < 48 * 64 ** 3 // 8                                                # (the previous generation kept 48 bytes per voxel)
    tex = ops.pack_nchw(scene["features"][1])
    w2c = torch.linalg.inv(scene["c2ws"]).contiguous()
    k1 = _scaled_intrinsics(scene["intrs"], 2)[1]
    gvol, vol = torch.zeros(8, 64, 64, 64, device="cuda"), torch.zeros(8, 64, 64, 64, device="cuda")
    cnt = torch.zeros(64 ** 3, device="cuda", dtype=torch.uint8)
    gf = torch.zeros_like(tex)
    scratch = torch.empty(need, device="cuda", dtype=torch.uint8)

    def call(dims, scr, nbytes, counts=(cnt,)):
        L.call("gens_volume_build_bwd_levels", L.ptr_table([tex]), L.int_table([240, 320]), L.int_table(dims), 1, L.ptr(w2c), L.ptr_table([k1]), 5, L.ptr_table([vol]),
               L.ptr_table(list(counts), torch.uint8), L.ptr_table([gvol]), L.ptr_table([gf]), scr, nbytes, L.stream())
    with pytest.raises(RuntimeError, match="scratch"):
        call([64], L.ptr(scratch, torch.uint8), need - 1)
    with pytest.raises(RuntimeError, match="multiple of 16"):
        call([24], L.ptr(scratch, torch.uint8), need)
    with pytest.raises(RuntimeError, match="null"):
        call([64], None, need)
    with pytest.raises(RuntimeError, match="null"):
        call([64], L.ptr(scratch, torch.uint8), need, counts=(None,))


@pytest.mark.parametrize("layout", ["planar", "packed"])
def test_k2_volume_scatter_lane_per_float_equals_lane_per_point(scene, layout, monkeypatch):
    """The volume gradients of both K2 backward passes (lookup_scatter_k: a lane per float, its own launch) against the scatter inside the
    per-point kernels (GENS_K2_SCATTER_PER_POINT), planar and packed gradients, 262 144 ray samples + points outside the cube, three levels:
    the same products in another order of the float atomics; d/dpts, gg_out and the second-order d/dpts are untouched (bit-identical)."""
    from gens_amd import lib as L, ops
    vols = scene["vols"][:3]
    lay = L.LAYOUT_PACKED if layout == "packed" else L.LAYOUT_PLANAR
    vs = ops.VolumeSet.packed(vols) if layout == "packed" else ops._vset(lay, vols)
    ro, rd = scene["rays_o"][:2048], scene["rays_d"][:2048]
    z = torch.linspace(0.2, 2.6, 128, device="cuda")
    pts = (ro[:, None] + rd[:, None] * z[None, :, None]).reshape(-1, 3).contiguous()
    n = pts.shape[0]
    g = torch.Generator(device="cuda").manual_seed(3)
    g_out = torch.randn(n, 12, device="cuda", generator=g)
    gg_pts = torch.randn(n, 3, device="cuda", generator=g)

    def run(per_point):
        if per_point:
            monkeypatch.setenv("GENS_K2_SCATTER_PER_POINT", "1")
        else:
            monkeypatch.delenv("GENS_K2_SCATTER_PER_POINT", raising=False)
        gv = [torch.zeros_like(t) for t in vs.tensors]
        g_pts = torch.empty(n, 3, device="cuda")
        L.call("gens_lookup_volume_bwd", vs.table, vs.dim_table, vs.n, lay, L.ptr(pts), L.ptr(g_out), n, L.ptr_table(gv), L.ptr(g_pts), L.stream())
        gv2 = [torch.zeros_like(t) for t in vs.tensors]
        gg_out, g_pts2 = torch.empty(n, 12, device="cuda"), torch.empty(n, 3, device="cuda")
        L.call("gens_lookup_volume_bwd2", vs.table, vs.dim_table, vs.n, lay, L.ptr(pts), L.ptr(g_out), L.ptr(gg_pts), None, n, L.ptr(gg_out),
               L.ptr_table(gv2), L.ptr(g_pts2), L.stream())
        return gv, gv2, (g_pts, gg_out, g_pts2)
    a1, a2, ar = run(False)
    b1, b2, br = run(True)
    for x, y in zip(ar, br):
        assert torch.equal(x, y)
    for x, y in zip(a1 + a2, b1 + b2):
        scale = float(y.abs().max())
        assert scale > 1.0 and float((x - y).abs().max()) <= 2e-6 * scale + 1e-6, (float((x - y).abs().max()), scale)
    only = [torch.zeros_like(t) if l == 1 else None for l, t in enumerate(vs.tensors)]           # a subset of the levels
    L.call("gens_lookup_volume_bwd", vs.table, vs.dim_table, vs.n, lay, L.ptr(pts), L.ptr(g_out), n, L.ptr_table(only), None, L.stream())
    assert float((only[1] - b1[1]).abs().max()) <= 2e-6 * float(b1[1].abs().max()) + 1e-6


def test_k4_feature_backward_at_step_size(scene):
    """The feature / image gradients of the source-view look-up (gens_lookup_feature_bwd: sixteen lanes per (point, view) pair, one float atomic
    per lane and level) at the size of a training step -- 61 003 points x 4 source views of the 480 x 640 five-level pyramid: against the CPU
    oracle's autograd on the first 8 192 points, and through size-independent properties on all of them (the two halves of the batch add up
    to the whole; doubling the cotangent doubles the result; zero cotangent rows add nothing)."""
    from gens_amd import ops
    from oracle import gens_oracle as K
    feats, imgs, intrs, c2ws = scene["features"], scene["imgs"], scene["intrs"], scene["c2ws"]
    g = torch.Generator().manual_seed(11)
    n = 61003
    pts = (torch.rand(n, 3, generator=g) * 1.6 - 0.8).cuda()
    cot = torch.randn(n, intrs.shape[0] - 1, 3 + 4 * len(feats), generator=g).cuda()

    def grads(p, c):
        fl = [f.detach().clone().requires_grad_(True) for f in feats]
        im = imgs.detach().clone().requires_grad_(True)
        fv, _, _ = ops.lookup_feature(p, ops.SceneViews(im, intrs, c2ws, fl))
        return torch.autograd.grad((fv * c).sum(), fl + [im])
    full = grads(pts, cot)
    m = 8192
    fl = [f.detach().cpu().clone().requires_grad_(True) for f in feats]
    im = imgs.detach().cpu().clone().requires_grad_(True)
    fv_ref, _, _ = K.lookup_feature(pts[:m].cpu(), im, intrs.cpu(), c2ws.cpu(), fl)
    ref = torch.autograd.grad((fv_ref * cot[:m].cpu()).sum(), fl + [im])
    for a, b in zip(grads(pts[:m].contiguous(), cot[:m].contiguous()), ref):
        scale = float(b.abs().max())
        # (a tap's weight moves with the projected coordinate: float32 round-off of 1e-4 pixel at x ~ 600 times a cotangent of ~4; g4 uses 1e-4 absolute)
        assert scale > 0.1 and float((a.cpu() - b).abs().max()) <= 2e-4 * scale, (float((a.cpu() - b).abs().max()), scale)
    h = n // 2 + 5
    first, second = grads(pts[:h].contiguous(), cot[:h].contiguous()), grads(pts[h:].contiguous(), cot[h:].contiguous())
    doubled = grads(pts, 2.0 * cot)
    masked = cot.clone()
    masked[h:] = 0.0
    only_first = grads(pts, masked)
    for a, p, q, d2, o in zip(full, first, second, doubled, only_first):
        scale = float(a.abs().max())
        assert float((p + q - a).abs().max()) <= 1e-5 * scale
        assert float((d2 - 2.0 * a).abs().max()) <= 1e-5 * scale
        assert float((o - p).abs().max()) <= 1e-5 * scale


# --------------------------------------------------------------------------------------------------- K17 / K18 at training-step size
def _train_case(n_levels, n, seed):
    from gens_amd import ops
    from oracle import sdf_train_oracle as T
    g = torch.Generator().manual_seed(seed)
    dims = [64, 32, 16, 8, 4][:n_levels]
    vols = [(0.3 * torch.randn(1, 4, d, d, d, generator=g)).cuda() for d in dims]
    W, b = T.shipped_weights(n_levels, seed=seed + 1, scale=1.0)
    pts = (torch.rand(n, 3, generator=g) * 2.2 - 1.1).cuda()
    cot = [torch.randn(n, k, generator=g).cuda() for k in (1, 3, 3)]
    return ops, [w.cuda() for w in W], [v.cuda() for v in b], vols, pts, cot


def _sdf_train_grads(ops, W, b, vols, pts, cot):
    Wd = [w.clone().requires_grad_(True) for w in W]
    bd = [v.clone().requires_grad_(True) for v in b]
    vd = [v.clone().requires_grad_(True) for v in vols]
    y, g, s = ops.SdfTrainStep(Wd, bd, vd, ops.VolumeSet.packed(vd))(pts)
    ((y * cot[0]).sum() + (g * cot[1]).sum() + (s * cot[2]).sum()).backward()
    return (y.detach(), g.detach(), s.detach()), [t.grad for t in Wd + bd + vd]


@pytest.mark.parametrize("n_levels", [3, 5])
def test_sdf_train_kernels_at_step_size_partition_and_order(n_levels):
    """K17 at the size of a training step (BASELINE config[2]: 512 rays x 128 samples + 1 024 random + 2 048 pseudo points = 68 608), where
    the oracle is too slow, through properties that do not depend on the size: (i) every point is evaluated independently of its
    neighbours in the batch -- a permutation of the points permutes y, g, s bit for bit; (ii) the parameter and volume gradients are
    sums over points -- the two halves of the batch add up to the whole (float32 summation order aside); (iii) the backward is linear in
    the cotangents."""
    n = 68608
    ops, W, b, vols, pts, cot = _train_case(n_levels, n, seed=70 + n_levels)
    (y, g, s), full = _sdf_train_grads(ops, W, b, vols, pts, cot)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1)).cuda()
    (yp, gp, sp), _ = _sdf_train_grads(ops, W, b, vols, pts[perm], [c[perm] for c in cot])
    assert torch.equal(yp, y[perm]) and torch.equal(gp, g[perm]) and torch.equal(sp, s[perm])
    h = n // 2 + 7                                                     # not a multiple of the 32-point tile
    _, first = _sdf_train_grads(ops, W, b, vols, pts[:h], [c[:h] for c in cot])
    _, second = _sdf_train_grads(ops, W, b, vols, pts[h:], [c[h:] for c in cot])
    for a, p, q in zip(full, first, second):
        scale = a.abs().max().clamp_min(1e-20)
        assert ((p + q - a).abs().max() / scale) < 2e-4
    _, doubled = _sdf_train_grads(ops, W, b, vols, pts, [2.0 * c for c in cot])
    for a, d2 in zip(full, doubled):
        # (doubling is exact term by term; the volume gradients are float atomics, summed in a different order on every launch: 68 608 points on
        # the 4^3 = 64 voxels of the coarsest level differ by up to 1.05e-5 of the largest entry between two runs)
        assert ((d2 - 2.0 * a).abs().max() / a.abs().max().clamp_min(1e-20)) < 5e-5
    assert all(torch.isfinite(t).all() for t in full)


def test_blend_train_kernels_at_step_size_partition():
    """K18 at training-step size (61 000 points x 4 source views of a 480 x 640 five-view scene): outputs are per point (a permutation
    permutes them bit for bit) and the parameter gradients of the two halves add up to the whole."""
    from gens_amd import ops, synthetic
    from gens_amd.models.modules.blending_network import BlendingNetwork
    sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=9)
    torch.manual_seed(2)
    net = BlendingNetwork(d_feature=20).cuda()
    views = ops.SceneViews(sc["imgs"].cuda(), sc["intrs"].cuda(), sc["c2ws"].cuda(), [f.cuda() for f in sc["features"]])
    g = torch.Generator().manual_seed(3)
    n = 61003
    pts = (torch.rand(n, 3, generator=g) * 1.6 - 0.8).cuda()
    cot = torch.randn(n, 3, generator=g).cuda()

    def run(p, c):
        net.zero_grad(set_to_none=True)
        rgb, vis = ops.blend_train(net, views, p)
        (rgb * c).sum().backward()
        return rgb.detach(), vis, [q.grad.clone() for q in net.parameters()]
    rgb, vis, full = run(pts, cot)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(4)).cuda()
    rgb_p, vis_p, _ = run(pts[perm], cot[perm])
    assert torch.equal(rgb_p, rgb[perm]) and torch.equal(vis_p, vis[perm])
    h = n // 2 + 3
    _, _, a = run(pts[:h], cot[:h])
    _, _, b = run(pts[h:], cot[h:])
    top = max(float(t.abs().max()) for t in full)
    for f, p, q in zip(full, a, b):
        assert float((p + q - f).abs().max()) < 2e-4 * max(float(f.abs().max()), 1e-3 * top)
    assert torch.isfinite(rgb).all() and vis.any()


def test_value_and_gradient_kernels_return_the_same_value_at_full_chunk_size():
    """K6t (gens_sdf_value) and K6g (gens_sdf_grad) share the forward chain instruction for instruction: at the size of one bench.py ray chunk
    (32 768 rays x 128 samples) the two values are the same float32 numbers, so the sampling passes and the render pass see ONE function; the
    gradient is orthogonal-free sanity-checked by a directional finite difference on the points where the SDF is smooth enough for float32."""
    from gens_amd import ops, synthetic
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.implicit_surface import ImplicitSurface
    dims = [256, 128, 64]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"]).to(dev).eval()
    vols = ops.VolumeSet.packed([v.to(dev) for v in synthetic.make_volumes(dims, seed=3)])
    n = 32768 * 128
    pts = torch.rand(n, 3, device=dev) * 2.2 - 1.1
    plan = ops.SdfMlpPlan(surf.sdf_network)
    value = ops.sdf_mlp(plan, vols, pts)
    sdf, grad = ops.sdf_mlp(plan, vols, pts, want_grad=True)
    assert torch.equal(value, sdf)
    assert torch.isfinite(grad).all()
    # directional derivative along a fixed direction on a subsample, central difference with h = 2e-3 (float32: ~1e-3 absolute noise)
    sub = pts[::4099].contiguous()
    d = torch.tensor([0.3, -0.5, 0.8], device=dev)
    d = d / d.norm()
    h = 2e-3
    fd = (ops.sdf_mlp(plan, vols, sub + h * d) - ops.sdf_mlp(plan, vols, sub - h * d))[:, 0] / (2 * h)
    an = (grad[::4099] * d).sum(-1)
    err = (fd - an).abs()
    assert err.median() < 5e-3 and (err < 0.05 * (1 + an.abs())).float().mean() > 0.97     # (kinks of the trilinear volumes and of softplus-100 excepted)


def test_split_half_kernels_at_full_chunk_size_agree_with_float32_and_with_each_other():
    """K6v (gens_sdf_value_f16) and K6gh (gens_sdf_grad_f16) at the size of one bench.py ray chunk (32 768 rays x 128 samples, volumes
    256 / 128 / 64): both stay within 1e-5 of the float32 value everywhere (their forward chains differ in the order of a layer's K
    blocks, so they are not bit-identical to each other), the split-half gradient within 1e-4 (1 + |grad|) of the float32 one, the
    directional finite difference of the split-half VALUE agrees with the split-half GRADIENT like the float32 pair does, the overflow
    flag stays clear, and the rows beyond the device-side count are not written."""
    from gens_amd import ops, synthetic
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.implicit_surface import ImplicitSurface
    dims = [256, 128, 64]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    surf = ImplicitSurface(gens_model_conf(volume_dims=tuple(dims))["implicit_surface"]).to(dev).eval()
    vols = ops.VolumeSet.packed([v.to(dev) for v in synthetic.make_volumes(dims, seed=3)])
    n = 32768 * 128
    pts = torch.rand(n, 3, device=dev) * 2.2 - 1.1
    plan = ops.SdfMlpPlan(surf.sdf_network)
    assert plan.value_ok and plan.grad_pieces is not None
    sdf32, grad32 = ops.sdf_mlp(plan, vols, pts, want_grad=True)
    value16 = ops.sdf_mlp(plan, vols, pts, precision="f16x2")
    sdf16, grad16 = ops.sdf_mlp(plan, vols, pts, want_grad=True, precision="f16x2")
    assert not plan.overflowed()
    assert (value16 - sdf32).abs().max() < 1e-5 and (sdf16 - sdf32).abs().max() < 1e-5
    assert torch.isfinite(grad16).all()
    assert ((grad16 - grad32).abs() <= 1e-4 * (1 + grad32.abs())).all()
    sub = pts[::4099].contiguous()
    d = torch.tensor([0.3, -0.5, 0.8], device=dev)
    d = d / d.norm()
    h = 2e-3
    fd = (ops.sdf_mlp(plan, vols, sub + h * d, precision="f16x2") - ops.sdf_mlp(plan, vols, sub - h * d, precision="f16x2"))[:, 0] / (2 * h)
    an = (grad16[::4099] * d).sum(-1)
    err = (fd - an).abs()
    assert err.median() < 5e-3 and (err < 0.05 * (1 + an.abs())).float().mean() > 0.97
    # index map + device-side count at this size: a ragged count in the middle of a 128-point tile
    idx = torch.randperm(n, device=dev)[:n // 2]
    count = torch.tensor([n // 2 - 12345], dtype=torch.int32, device=dev)
    s_out, g_out = torch.full((n, 1), 100.0, device=dev), torch.full((n, 3), -7.0, device=dev)
    ops.sdf_mlp(plan, vols, pts, index=idx, want_grad=True, sdf_out=s_out, grad_out=g_out, precision="f16x2", count=count)
    live = idx[:n // 2 - 12345]
    assert torch.equal(s_out[live], sdf16[live]) and torch.equal(g_out[live], grad16[live])      # the same numbers whatever tile a point lands in
    untouched = torch.ones(n, dtype=torch.bool, device=dev)
    untouched[live] = False
    assert bool((s_out[untouched] == 100.0).all()) and bool((g_out[untouched] == -7.0).all())
    assert not plan.overflowed()


@pytest.mark.parametrize("n_levels,n_points", [(1, 1001), (2, 4097), (3, 65537), (4, 333), (5, 20001)])
def test_k2_forward_lane_pairs_equal_the_lane_per_item_kernel(scene, n_levels, n_points, monkeypatch):
    """The packed forward reads with two lanes to a texel line (lookup_fwd_paired_k); the lane-per-item kernel (GENS_K2_NO_PAIRS) is the
    reference of the goldens g2: bit-identical for every level count, odd item counts (a pair straddles two points / the end of the
    launch), points on and outside the faces of the volume, and not-a-number points."""
    from gens_amd import ops
    g = torch.Generator(device="cpu").manual_seed(n_points)
    pts = (torch.rand(n_points, 3, generator=g) * 2.6 - 1.3).cuda()                 # a fifth of them outside [-1, 1]^3
    pts[::97] = torch.tensor([1.0, -1.0, 1.0], device="cuda")                      # corners
    pts[5::89, 1] = 1.0                                                              # on a face
    pts[7::101, 2] = float("nan")
    pts[11::103] = 1e30
    vset = ops.VolumeSet.packed(scene["vols"][:n_levels])
    new = ops.lookup_volume(pts, vset)
    monkeypatch.setenv("GENS_K2_NO_PAIRS", "1")
    old = ops.lookup_volume(pts, vset)
    assert new.shape == (n_points, 4 * n_levels)
    assert torch.equal(new.view(torch.int32), old.view(torch.int32))
    assert bool(torch.isfinite(new[torch.isfinite(pts).all(1)]).all())          # incl. the points at 1e30: no 0 * inf from overflowing weights
    assert float(new[11::103].abs().max()) == 0.0                               # F.grid_sample skips out-of-bounds taps: exactly zero


@pytest.mark.parametrize("n_src,n_feat,n_points", [(4, 5, 30011), (3, 5, 7777), (2, 3, 12345), (1, 3, 501), (4, 2, 999)])
def test_k4_forward_lane_pairs_equal_the_lane_per_item_kernel(scene, n_src, n_feat, n_points, monkeypatch):
    """lookup_feature's forward with the bilinear taps read by lane pairs (sample_texel_pair) against the lane-per-item read
    (GENS_K4_NO_PAIRS): bit-identical rows, ray differences and visibility flags for 1 - 4 source views (pairs straddle points for odd
    counts), 2 / 3 / 5 feature levels, points behind the cameras and far outside every frustum."""
    from gens_amd import ops
    g = torch.Generator(device="cpu").manual_seed(n_points)
    pts = (torch.rand(n_points, 3, generator=g) * 2.4 - 1.2).cuda()
    pts[::53] *= 40.0                                                                # far outside / behind some cameras
    nv = n_src + 1
    views = ops.SceneViews(scene["imgs"][:nv].contiguous(), scene["intrs"][:nv].contiguous(), scene["c2ws"][:nv].contiguous(),
                           [f[:nv].contiguous() for f in scene["features"][:n_feat]])
    new = ops.lookup_feature(pts, views)
    monkeypatch.setenv("GENS_K4_NO_PAIRS", "1")
    old = ops.lookup_feature(pts, views)
    monkeypatch.setenv("GENS_K4_NO_UNROLL", "1")
    rolled = ops.lookup_feature(pts, views)
    for a, b, c in zip(new, old, rolled):
        assert a.shape == b.shape
        if a.dtype == torch.bool:
            assert torch.equal(a, b) and torch.equal(a, c)
        else:
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and torch.equal(a.view(torch.int32), c.view(torch.int32))
    assert 0 < int(new[2].sum()) < new[2].numel()
