"""K19, the step-boundary kernels of a training step (gens_amd/csrc/k19_step.hip), each against the PyTorch operations it replaces."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_scene_setup_matches_torch_inverse():
    """gens_scene_setup against torch.inverse in float64 (volume.py:34, projector.py:322,364, implicit_surface.py:242): the float64
    Gauss-Jordan result rounded to float32 is within one float32 ulp of the exact inverse; per-level intrinsics are exact."""
    from gens_amd import ops, synthetic
    for nv in (1, 3, 5, 11):
        intrs, c2ws, _, _ = synthetic.make_cameras(nv, 480, 640)
        g = torch.Generator().manual_seed(nv)
        c2ws[:, :3, 3] += 0.1 * torch.randn(nv, 3, generator=g)
        cams = ops.SceneCams(intrs.cuda(), c2ws.cuda())
        want = torch.linalg.inv(c2ws.double())
        assert (cams.w2c.cpu().double() - want).abs().max() <= 1e-7 * want.abs().max()
        for lvl in range(8):
            k = intrs.clone()
            k[:, :2] *= 0.5 ** lvl
            assert torch.equal(cams.ks[lvl].cpu(), k)
        assert (cams.rot_inv.cpu().view(3, 3).double() - torch.linalg.inv(c2ws[0, :3, :3].double())).abs().max() <= 1e-7
        kinv = torch.linalg.inv(intrs.double())[0, :3, :3]
        assert ((cams.kinv_ref.cpu().double() - kinv).abs() <= 1e-7 * kinv.abs().max()).all()
        assert int(cams.status.item()) == 0
        cams.check()
    # the instance is shared by the volume build and the renderer of one forward pass, and rebuilt when a tensor changes
    i_d, c_d = intrs.cuda(), c2ws.cuda()
    a = ops.SceneCams.of(i_d, c_d)
    assert ops.SceneCams.of(i_d, c_d) is a
    c_d[0, 0, 3] += 1.0
    assert ops.SceneCams.of(i_d, c_d) is not a
    bad = c2ws.clone()
    bad[1] = 0.0
    cams = ops.SceneCams(intrs.cuda(), bad.cuda())
    assert int(cams.status.item()) == 1 and torch.isnan(cams.w2c[1]).all() and torch.isfinite(cams.w2c[0]).all()
    with pytest.raises(RuntimeError, match="singular"):
        cams.check()


def test_pack_maps_equals_pack_nchw_and_its_adjoint():
    from gens_amd import ops
    g = torch.Generator().manual_seed(0)
    shapes = [(3, 3, 48, 64), (3, 4, 48, 64), (3, 4, 24, 32), (3, 4, 12, 16), (3, 5, 6, 8), (2, 12, 7, 9)]
    maps = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    texs = ops.pack_maps(maps)
    for m, t in zip(maps, texs):
        assert torch.equal(t, ops.pack_nchw(m.detach()))
    assert ops.pack_maps(maps)[0] is texs[0]                     # kept on the tensor
    cots = [torch.randn(t.shape, generator=g).cuda() for t in texs]
    cots[2] = None
    torch.autograd.backward([t for t, c in zip(texs, cots) if c is not None], [c for c in cots if c is not None])
    for m, c in zip(maps, cots):
        if c is None:
            assert m.grad is None
        else:
            assert torch.equal(m.grad, c[..., :m.shape[1]].permute(0, 3, 1, 2))
    with torch.no_grad():
        maps[0].add_(1.0)                                         # a new version: packed again
    assert not torch.equal(ops.pack_maps(maps)[0], texs[0])


@pytest.mark.parametrize("n0,n1,n2,p0,p2", [(65536, 1024, 2048, 0.9, 0.9), (1000, 0, 0, 0.5, 0.0), (4096, 1024, 300, 0.0, 0.5), (7, 3, 5, 0.0, 0.0),
                                             (0, 16, 0, 0.0, 0.0), (70000, 1024, 2048, 0.01, 1.0)])
def test_compact_points_against_nonzero(n0, n1, n2, p0, p2):
    """gens_compact_points against the reference's selection (implicit_surface.py:121-124,174-177: nonzero with the first-ten rescue) on
    the ray samples, the always-selected random points and the pseudo points' own nonzero (:484-497)."""
    from gens_amd import lib as L
    g = torch.Generator().manual_seed(n0 + n2)
    n = n0 + n1 + n2
    valid = torch.cat([torch.rand(n0, generator=g) < p0, torch.ones(n1, dtype=torch.bool), torch.rand(n2, generator=g) < p2]).to(torch.uint8)
    idx = torch.full((max(n, 1),), -1, dtype=torch.int64, device="cuda")
    counts = torch.zeros(3, dtype=torch.int32, device="cuda")
    y = torch.full((max(n, 1),), 7.0, device="cuda")
    g3 = torch.full((max(n, 1), 3), 7.0, device="cuda")
    rgb = torch.full((max(n0, 1), 3), 7.0, device="cuda")
    vis = torch.full((max(n0, 1), 4), 7, dtype=torch.uint8, device="cuda")
    z = torch.rand(max(n0, 1), generator=g).cuda() * 3 - 1
    var = torch.tensor([0.3], device="cuda")
    scal = torch.zeros(4, device="cuda")
    valid_d = valid.cuda()                                                      # (kept alive: a temporary's memory would be handed to the scratch buffer)
    scratch = torch.empty(L.load().gens_compact_points_scratch(n), dtype=torch.int32, device="cuda")
    L.call("gens_compact_points", L.ptr(valid_d, torch.uint8), n0, n1, n, L.ptr(idx, torch.int64), L.ptr(counts, torch.int32), L.ptr(y),
           L.ptr(g3), None, L.ptr(rgb), L.ptr(vis, torch.uint8), 4, L.ptr(z), z.numel(), L.ptr(var), L.ptr(scal), L.ptr(scratch, torch.int32), L.stream())
    ray = torch.nonzero(valid[:n0])[:, 0]
    if ray.numel() < 1:
        ray = torch.arange(min(10, n0))
    pseudo = torch.nonzero(valid[n0 + n1:])[:, 0] + n0 + n1
    want = torch.cat([ray, torch.arange(n0, n0 + n1), pseudo])
    assert counts.tolist() == [want.numel(), ray.numel(), pseudo.numel()]
    assert torch.equal(idx[:want.numel()].cpu(), want)
    sel = torch.zeros(max(n, 1), dtype=torch.bool)
    sel[want] = True
    y_want = torch.where(sel, torch.tensor(7.0), torch.where(torch.arange(max(n, 1)) < n0, torch.tensor(100.0), torch.tensor(0.0)))[:n]
    assert torch.equal(y.cpu()[:n], y_want) and torch.equal(g3.cpu()[:n], torch.where(sel[:n, None], torch.tensor(7.0), torch.tensor(0.0)).expand(n, 3))
    assert torch.equal(rgb.cpu()[:n0], torch.where(sel[:n0, None], torch.tensor(7.0), torch.tensor(0.0)).expand(n0, 3))
    assert torch.equal(vis.cpu()[:n0], torch.where(sel[:n0, None], torch.tensor(7), torch.tensor(0)).to(torch.uint8).expand(n0, 4))
    inv_s = float(torch.exp(var * 10.0).clip(1e-6, 1e6))
    assert float(scal[0]) == float(z.max()) and abs(float(scal[1]) - inv_s) <= 2e-6 * inv_s and abs(float(scal[2]) * inv_s - 1) < 1e-5 and float(scal[3]) == 1.0


@pytest.mark.parametrize("dims", [(64, 32, 16), (32, 16, 8, 8, 4), (24,)])
def test_tv_levels_equal_the_per_level_kernels(dims):
    """One launch for all levels + one finishing workgroup against the per-level kernels and their torch epilogue (value and gradients)."""
    from gens_amd import ops
    g = torch.Generator().manual_seed(len(dims))
    vols = [torch.randn(1, 4, d, d, d, generator=g).cuda().requires_grad_(True) for d in dims]
    masks = [(torch.rand(1, 1, d, d, d, generator=g) > 0.3).float().cuda() for d in dims]
    assert ops.tv_levels_ok(vols, masks)
    tv = ops.tv_regularization(vols, masks)
    (3.0 * tv).backward()
    got = [v.grad.clone() for v in vols]
    for v in vols:
        v.grad = None
    ref = 0
    for lvl, (v, m) in enumerate(zip(vols, masks)):
        ref = ref + ops._TVLevel.apply(v, m) * 0.5 ** lvl
    (3.0 * ref).backward()
    assert abs(float(tv) - float(ref)) <= 2e-6 * abs(float(ref))
    for a, v in zip(got, vols):
        assert (a - v.grad).abs().max() <= 2e-6 * v.grad.abs().max()
    # odd sizes keep the per-level path
    odd = [torch.randn(1, 4, 6, 6, 6, generator=g).cuda()]
    assert not ops.tv_levels_ok(odd, [torch.ones(1, 1, 6, 6, 6).cuda()])
    assert torch.isfinite(ops.tv_regularization(odd, [torch.ones(1, 1, 6, 6, 6).cuda()]))


@pytest.mark.parametrize("nv", [3, 5])
def test_fused_patch_warp_equals_the_operator_by_operator_mirror(nv):
    """gens_patch_warp_fwd / _bwd against projector.surface_patch_warp (the reference's tensor pipeline on the K9 reads, pinned to the
    reference by golden g7): the same patches, and the same gradient with respect to the crossing depth."""
    from gens_amd import ops, synthetic
    from gens_amd.models.modules.projector import surface_patch_warp
    sc = synthetic.make_scene(nv=nv, h=96, w=128, n_levels=3, seed=nv)
    g = torch.Generator().manual_seed(1)
    b = 70
    pix = torch.stack([torch.randint(20, 108, (b,), generator=g), torch.randint(20, 76, (b,), generator=g)], -1)
    ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 96, 128, pixels=pix)
    ro, rd = ro.cuda(), rd.cuda()
    intrs, c2ws = sc["intrs"].cuda(), sc["c2ws"].cuda()
    feats = [f.cuda() for f in sc["features"]]
    warp = ops.build_warp_features(feats[:3])
    z = (1.6 + 1.2 * torch.rand(b, generator=g)).cuda().requires_grad_(True)
    g0 = torch.randn(b, 3, generator=g).cuda()
    g0[3] = 0.0                                                     # a zero gradient: the 1e-8 floor of :308-309
    cams = ops.SceneCams.of(intrs, c2ws)
    ref, src = ops.patch_warp(z, ro, rd, g0, cams, warp)
    cot = torch.randn(src.shape, generator=g).cuda()
    (src * cot).sum().backward()
    gz = z.grad.clone()
    z.grad = None
    # the mirror of the reference's pipeline
    pts = ro[:, None, :] + rd[:, None, :] * z[:, None, None]
    n = g0.reshape(b, 1, 3)
    nn = torch.linalg.norm(n, dim=-1, keepdim=True)
    n = n / torch.where(nn <= 0, torch.full_like(nn, 1e-8), nn)
    normals = (n @ c2ws[0, :3, :3]).detach()
    ref_t, src_t = surface_patch_warp(pts, normals, warp, intrs, c2ws)
    (src_t * cot).sum().backward()
    assert ref.shape == ref_t.shape == (1, b, 121, 12) and src.shape == src_t.shape == (nv - 1, b, 121, 12)
    assert (ref - ref_t).abs().max() < 1e-4
    # bilinear reads of white-noise features: a 1e-4 px difference of the float32 homography chain moves a sample by ~1e-4
    assert (src - src_t).abs().mean() < 2e-4 and (src - src_t).abs().max() < 2e-2
    # (a sample within 1e-4 px of a texel boundary reads its x / y derivative from the neighbouring cell in one of the two float32 chains:
    # with white-noise features that is one O(1) term among a ray's 484; the rays are held to 2e-3 of the largest gradient, all but a few)
    scale = float(z.grad.abs().max())
    err = (gz - z.grad).abs()
    assert float((err < 2e-3 * scale).float().mean()) >= 0.9 and float(err.max()) < 3e-2 * scale, (err.max(), scale)
