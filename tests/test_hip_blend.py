"""GPU parity of the fused look-up + blending kernel (gens_blend_views) against K4 + the PyTorch BlendingNetwork."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(nv, n_levels, seed, n):
    from gens_amd import ops, synthetic
    from gens_amd.models.modules.blending_network import BlendingNetwork
    sc = synthetic.make_scene(nv=nv, h=48, w=64, n_levels=n_levels, seed=seed)
    torch.manual_seed(seed)
    net = BlendingNetwork(d_feature=4 * n_levels).cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    views = ops.SceneViews(sc["imgs"].cuda(), sc["intrs"].cuda(), sc["c2ws"].cuda(), [f.cuda() for f in sc["features"]])
    g = torch.Generator().manual_seed(seed + 1)
    pts = (torch.rand(n, 3, generator=g) * 2 - 1)
    pts[:3] = torch.tensor([[0, 0, -3.0], [2.5, 0, -2.0], [0.9, 0.9, 0.9]])[:min(3, n)]    # behind / outside some views
    return ops, net, views, pts.cuda()


@pytest.mark.parametrize("nv,n_levels,n", [(5, 5, 1000), (3, 5, 333), (4, 5, 250), (5, 3, 64), (5, 5, 1), (5, 1, 130), (5, 2, 77), (3, 1, 130)])
def test_fused_blend_matches_k4_plus_torch_network(nv, n_levels, n):
    ops, net, views, pts = _setup(nv, n_levels, seed=nv * 10 + n_levels, n=n)
    fv, rd, mk = ops.lookup_feature(pts, views)
    with torch.no_grad():
        ref = net(fv, rd, mk)
    rgb, vis = ops.blend_views(ops.BlendPlan(net), views, pts)
    assert torch.equal(vis.bool(), mk)
    assert (rgb - ref).abs().max() < 2e-5, (rgb - ref).abs().max()


def test_fused_blend_indexed_scatter():
    ops, net, views, pts = _setup(5, 5, seed=3, n=400)
    idx = torch.nonzero(torch.rand(400, generator=torch.Generator().manual_seed(0)) > 0.5)[:, 0].cuda()
    dense_rgb, dense_vis = ops.blend_views(ops.BlendPlan(net), views, pts)
    rgb, vis = ops.blend_views(ops.BlendPlan(net), views, pts, index=idx)
    keep = torch.zeros(400, dtype=torch.bool, device="cuda")
    keep[idx] = True
    assert torch.allclose(rgb[keep], dense_rgb[keep], atol=1e-6) and torch.equal(vis[keep], dense_vis[keep])
    assert (rgb[~keep] == 0).all() and (vis[~keep] == 0).all()


@pytest.mark.parametrize("nv,n", [(5, 600), (3, 200), (4, 200)])
def test_fused_blend_matches_the_cpu_oracle(nv, n):
    """gens_blend_views against the CPU oracle directly (oracle.gens_oracle.lookup_feature + oracle.render_oracle.blend_mlp, themselves
    pinned to the reference's lookup_feature / BlendingNetwork by goldens g4 and g9a-c): S = 4 source views is the shipped count
    (confs/gens.conf:13), S = 2 the validation one.  Not a self-comparison: nothing of the device path is on the reference side."""
    from oracle import gens_oracle as K
    from oracle import render_oracle as R
    from gens_amd import synthetic
    ops, net, views, pts = _setup(nv, 5, seed=77 + nv, n=n)
    sc = synthetic.make_scene(nv=nv, h=48, w=64, n_levels=5, seed=77 + nv)
    sd = {"color_network." + k: v.detach().cpu() for k, v in net.state_dict().items()}
    fv, rd, mk = K.lookup_feature(pts.cpu(), sc["imgs"], sc["intrs"], sc["c2ws"], sc["features"])
    live = mk.any(1)                                        # a point no source view sees: softmax over -1e9 only, colour is arbitrary
    ref = R.blend_mlp(sd, torch.nan_to_num(fv), torch.nan_to_num(rd), mk)
    rgb, vis = ops.blend_views(ops.BlendPlan(net), views, pts)
    assert torch.equal(vis.bool().cpu(), mk)
    err = (rgb.cpu() - ref)[live].abs().max()
    assert err < 2e-5, err


@pytest.mark.parametrize("nv,n_levels,n", [(5, 5, 203), (3, 5, 50), (4, 3, 64), (5, 5, 5)])
def test_training_kernels_match_the_cpu_oracle(nv, n_levels, n):
    """K18 (gens_blend_train_fwd / _bwd + the batched weight-gradient products + K4's backward for the maps) against autograd over the
    CPU oracle (oracle.gens_oracle.lookup_feature + oracle.render_oracle.blend_mlp): colours, and the gradient of a random linear
    functional of them with respect to every parameter of the network, the feature pyramid and the images."""
    from oracle import gens_oracle as K
    from oracle import render_oracle as R
    from gens_amd import synthetic
    ops, net, views, pts = _setup(nv, n_levels, seed=50 + nv, n=n)
    sc = synthetic.make_scene(nv=nv, h=48, w=64, n_levels=n_levels, seed=50 + nv)
    g = torch.Generator().manual_seed(3)
    cot = torch.randn(n, 3, generator=g)
    # oracle
    sd = {"color_network." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    feats_c = [f.clone().requires_grad_(True) for f in sc["features"]]
    imgs_c = sc["imgs"].clone().requires_grad_(True)
    fv, rd, mk = K.lookup_feature(pts.cpu(), imgs_c, sc["intrs"], sc["c2ws"], feats_c)
    ref = R.blend_mlp(sd, torch.nan_to_num(fv), torch.nan_to_num(rd), mk)
    live = mk.any(1)
    (ref[live] * cot[live]).sum().backward()
    # device: maps with gradients
    feats_d = [f.cuda().requires_grad_(True) for f in sc["features"]]
    imgs_d = sc["imgs"].cuda().requires_grad_(True)
    views = ops.SceneViews(imgs_d, sc["intrs"].cuda(), sc["c2ws"].cuda(), feats_d)
    rgb, vis = ops.blend_train(net, views, pts)
    assert torch.equal(vis.cpu(), mk)
    assert (rgb.cpu() - ref.detach())[live].abs().max() < 3e-5
    (rgb[live.cuda()] * cot.cuda()[live.cuda()]).sum().backward()
    worst = {}
    top = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for k, p in net.named_parameters():
        r = sd["color_network." + k].grad
        # (the bias in front of the softmax has an analytically zero gradient -- 1e-8 of round-off on both sides: the floor keeps it out)
        worst[k] = float((p.grad.cpu() - r).abs().max()) / max(float(r.abs().max()), 1e-3 * top)
    for i in range(n_levels):
        worst[f"feat{i}"] = float((feats_d[i].grad.cpu() - feats_c[i].grad).abs().max()) / max(float(feats_c[i].grad.abs().max()), 1e-6)
    worst["imgs"] = float((imgs_d.grad.cpu() - imgs_c.grad).abs().max()) / max(float(imgs_c.grad.abs().max()), 1e-6)
    print({k: f"{v:.1e}" for k, v in worst.items()})
    print("s:", float(net.s.grad), float(sd["color_network.s"].grad), "largest parameter gradient", top)
    # the anti-alias temperature's gradient is a sum of cancelling per-sample terms, three to five orders of magnitude below the other
    # gradients: its float32 round-off is judged against THEIR scale
    worst.pop("s")
    assert abs(float(net.s.grad) - float(sd["color_network.s"].grad)) < 3e-5 * top
    bad = {k: v for k, v in worst.items() if v >= 2e-3}
    assert not bad, (bad, {k: float(sd["color_network." + k].grad.abs().max()) for k in bad if "color_network." + k in sd}, top)


@pytest.mark.parametrize("nv,n_levels,n", [(5, 5, 1), (5, 5, 15), (5, 5, 17), (5, 3, 1000), (5, 1, 130), (5, 2, 64), (5, 4, 333),
                                           (3, 5, 1), (3, 5, 31), (3, 5, 33), (3, 3, 1000), (3, 1, 130), (3, 2, 64), (3, 4, 333),
                                           (4, 5, 1), (4, 5, 15), (4, 5, 17), (4, 3, 1000), (4, 1, 130)])
def test_transposed_blend_kernel_equals_row_major_kernel(nv, n_levels, n, monkeypatch):
    """gens_blend_views_t (two to four source views -- the shipped counts, confs/gens.conf:9,24, confs/gens_finetune.conf:15: 64 rows per
    wave, activations in registers in quad layout, mean / variance columns once per point; three views with a dead fourth lane per point)
    against gens_blend_views (32 rows per wave through an LDS tile): the same float32 MFMA products in another order; with an index
    map and a device-side count as implicit_surface.py:196-199 passes them, untouched outputs keep their fill values."""
    ops, net, views, pts = _setup(nv, n_levels, seed=70 + n_levels, n=n)
    g = torch.Generator().manual_seed(n)
    idx = torch.randperm(n, generator=g).cuda()
    count = torch.tensor([max(1, (3 * n) // 4)], dtype=torch.int32, device="cuda")
    plan = ops.BlendPlan(net)

    def run():
        rgb = torch.full((n, 3), -7.0, device="cuda")
        vis = torch.full((n, nv - 1), 9, dtype=torch.uint8, device="cuda")
        ops.blend_views(plan, views, pts, index=idx, rgb_out=rgb, vis_out=vis, count=count)
        return rgb, vis

    new = run()
    monkeypatch.setattr(ops.kernels, "blend", "rowmajor")
    old = run()
    live, dead = idx[:int(count)], idx[int(count):]
    assert torch.equal(new[1], old[1])
    assert (new[0][live] - old[0][live]).abs().max() < 5e-6
    assert (new[0][dead] == -7).all() and (new[1][dead] == 9).all()


def test_transposed_blend_kernel_rejects_other_view_counts():
    from gens_amd import lib as L
    ops, net, views, pts = _setup(5, 5, seed=3, n=8)
    plan = ops.BlendPlan(net)
    feats = [ops.aligned16(f.detach()) for f in views.feat_tex]
    hw = [d for f in views.feat_tex for d in f.shape[1:3]]
    out = torch.zeros(8, 3, device="cuda")
    for entry, nv, msg in (("gens_blend_views4", 4, "four source views"), ("gens_blend_views_t", 2, "two to four source views"),
                           ("gens_blend_views_t", 6, "two to four source views")):
        with pytest.raises(RuntimeError, match=msg):
            L.call(entry, L.ptr_table(feats, align=16), L.int_table(hw), 5, L.ptr(ops.aligned16(views.imgs_tex.detach()), align=16),
                   L.ptr(views.w2c), L.ptr(views.intr), L.ptr(views.c2w), nv, L.ptr(plan.t_stream), L.ptr(plan.t_tab), plan.scalars, L.ptr(pts), None, 8,
                   None, L.ptr(out), None, L.stream())


def test_six_source_views_keep_the_row_major_kernel():
    """More than four source views (the C ABI allows up to 16): gens_blend_views, against K4 + the PyTorch network."""
    ops, net, views, pts = _setup(7, 3, seed=5, n=200)
    fv, rd, mk = ops.lookup_feature(pts, views)
    with torch.no_grad():
        ref = net(fv, rd, mk)
    rgb, vis = ops.blend_views(ops.BlendPlan(net), views, pts)
    assert torch.equal(vis.bool(), mk) and (rgb - ref).abs().max() < 2e-5


@pytest.mark.parametrize("rowmajor,nv", [(False, 5), (True, 5), (False, 3), (False, 4)])
def test_blend_kernels_propagate_not_a_number_inputs(rowmajor, nv, monkeypatch):
    """A NaN feature texel must give a NaN colour exactly for the points whose rows read it, as through the PyTorch layers (the median form
    of the ELU would drop it: the kernels carry a poison term in the row mask)."""
    ops, net, views, pts = _setup(nv, 5, seed=9, n=3000)
    views.feat_tex[1][2, 6:18, 8:24, 1] = float("nan")             # view 2, level 1, channel 1
    fv, rd, mk = ops.lookup_feature(pts, views)
    with torch.no_grad():
        ref = net(fv, rd, mk)
    bad = torch.isnan(ref).any(1)
    assert 0 < int(bad.sum()) < 2500
    if rowmajor:
        monkeypatch.setattr(ops.kernels, "blend", "rowmajor")
    rgb, vis = ops.blend_views(ops.BlendPlan(net), views, pts)
    assert torch.equal(torch.isnan(rgb).any(1), bad)
    # blending_network.py:93-95: weight = (e - min e) * mask / (sum + 1e-8).  With two source views the smaller e is the minimum, so the sum
    # is ONE difference of two exponentials: where the two viewing angles agree to ~1e-4 the reference's own float32 weights are noise (one
    # ulp of either exponential moves them by per cent), so such points are held to finiteness only
    e = torch.exp(net.s.detach().abs() * (rd[..., 3] - 1))
    wsum = ((e - e.min(dim=1, keepdim=True)[0]) * mk).sum(1)
    firm = ~bad & ((wsum > 1e-4) | (wsum == 0))          # d w / d e = 1e-8 / (sum + 1e-8)^2 <= 1 there
    assert int(firm.sum()) > 0.5 * int((~bad).sum())
    assert (rgb[firm] - ref[firm]).abs().max() < 2e-5
    assert torch.isfinite(rgb[~bad]).all()


@pytest.mark.parametrize("nv,n_levels", [(5, 5), (3, 5), (4, 3), (5, 1)])
def test_device_side_stream_packing_equals_the_host_packing(nv, n_levels):
    """gens_blend_pack_t (one launch from the raw parameters: the training step's forward) against gens_amd.ops._pack_blend_t, bit for bit;
    and the training forward through the transposed kernel against the row-major training kernel."""
    from gens_amd import lib as L
    ops, net, views, pts = _setup(nv, n_levels, seed=33 + nv, n=257)
    plan = ops.BlendPlan(net)
    w = [p.detach().reshape(-1).contiguous() if p.dim() == 0 else p.detach().contiguous() for p in ops.blend_params(net)]
    groups = L.load().gens_blend_views_t_groups(n_levels)
    wstream = torch.empty((groups + 2) * 64 * 4, device="cuda")
    tab, sc = torch.empty(320, device="cuda"), torch.empty(4, device="cuda")
    L.call("gens_blend_pack_t", L.ptr_table(w), n_levels, L.ptr(wstream, align=16), L.ptr(tab), L.ptr(sc), L.stream())
    assert torch.equal(wstream.view(-1, 64, 4), plan.t_stream)
    assert torch.equal(tab.view(10, 4, 8), plan.t_tab)
    assert torch.allclose(sc.cpu(), torch.tensor(list(plan.scalars)))
    rgb, vis = ops.blend_train(net, views, pts)
    ops.kernels.blend_train_fwd = "rowmajor"
    try:
        rgb_r, vis_r = ops.blend_train(net, views, pts)
    finally:
        ops.kernels.blend_train_fwd = "transposed"
    assert torch.equal(vis, vis_r) and (rgb - rgb_r).abs().max() < 3e-5      # (hardware exp / rcp in the transposed kernel; two views: ill-conditioned weights)


@pytest.mark.parametrize("nv,n_levels,n", [(5, 5, 20011), (3, 5, 777), (4, 3, 4099), (5, 1, 33)])
def test_weight_gradient_sums_inside_the_backward_launch_equal_operand_rows_plus_k14(nv, n_levels, n):
    """gens_blend_train_bwd_acc (persistent workgroups, the eleven [dW | db] blocks summed on the matrix cores out of LDS, one block per workgroup,
    a fixed-order reduction) against the operand-row form (gens_blend_train_bwd + gens_gemm_tn_batch): every parameter gradient of the network,
    with more row tiles than workgroups (20 011 points: 2 502 tiles on 512 workgroups) and a last tile that is partly empty."""
    from gens_amd.ops import base
    res = {}
    saved = base.kernels.blend_train_wgrad
    try:
        for mode in ("rows", "inside"):
            ops, net, views, pts = _setup(nv, n_levels, seed=70 + nv, n=n)
            base.kernels.blend_train_wgrad = mode
            cot = torch.randn(n, 3, generator=torch.Generator().manual_seed(4)).cuda()
            rgb, _ = ops.blend_train(net, views, pts)
            (rgb * cot).sum().backward()
            res[mode] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    finally:
        base.kernels.blend_train_wgrad = saved
    top = max(float(g.abs().max()) for g in res["rows"].values())
    for k, a in res["rows"].items():
        b = res["inside"][k]
        assert bool(torch.isfinite(b).all()), k
        # (the anti-alias temperature's gradient is a sum of cancelling per-sample terms, each divided by a sum of view weights that may be ~1e-6: it is
        # judged against the other gradients' scale, as in the oracle test; the bias in front of the soft-max has an analytically ZERO gradient --
        # sum_v score_bar_v = 0 per point -- 1e-7 of round-off in either form)
        if k in ("s", "rgb_fc.4.bias"):
            assert float((a - b).abs().max()) <= 3e-5 * top, (k, float((a - b).abs().max()), top)
            continue
        assert float((a - b).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-2 * top), (k, float((a - b).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize("nv,n_levels,n", [(5, 3, 1000), (5, 5, 203), (3, 5, 77), (4, 3, 130), (4, 5, 33), (5, 1, 64), (5, 2, 50), (5, 4, 41), (3, 3, 5000)])
def test_transposed_backward_equals_the_row_major_backward_layer_by_layer(nv, n_levels, n):
    """gens_blend_train_bwd_t (round 6: a wave per 16 rows, weights in LDS, activations from layer to layer in registers) against
    gens_blend_train_bwd (32-row workgroups through LDS tiles) on the same points, with an index map and a device-side count as
    implicit_surface.py:196-199 passes them: the operand rows of EVERY layer -- [input | 1] and the cotangent of the pre-activation, which is the
    whole forward and the whole reverse chain --, the cotangent of the looked-up rows, d loss / d |s|, and the eleven [dW | db] blocks against the
    batched products of the row-major kernel's operand rows."""
    from gens_amd import lib as L
    ops, net, views, pts = _setup(nv, n_levels, seed=90 + nv + n_levels, n=n)
    dev = pts.device
    s, f = nv - 1, 3 + 4 * n_levels
    g = torch.Generator().manual_seed(n)
    idx = torch.randperm(n, generator=g).cuda()
    n_live = max(1, (3 * n) // 4)
    count = torch.tensor([n_live], dtype=torch.int32, device=dev)
    g_rgb = torch.randn(n, 3, generator=g).cuda()
    w = [p.detach().reshape(-1).contiguous() if p.dim() == 0 else p.detach().contiguous() for p in ops.blend_params(net)]
    feats = [ops.aligned16(t.detach()) for t in views.feat_tex]
    imgs = ops.aligned16(views.imgs_tex.detach())
    hw = [d for t in feats for d in t.shape[1:3]]
    args = (L.ptr_table(feats, align=16), L.int_table(hw), n_levels, L.ptr(imgs, align=16), L.ptr(views.w2c), L.ptr(views.intr), L.ptr(views.c2w), nv,
            L.ptr_table(w), L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(g_rgb))
    ins = [4, 16, 3 * f, 64, 32, 32, 32, 32, 37, 16, 8]
    outs = [16, f, 64, 32, 32, 33, 32, 1, 16, 8, 1]
    ev = lambda x: (x + 1) // 2 * 2  # noqa: E731
    lib = L.load()
    # row-major: operand rows + the batched products
    rows_a = lib.gens_blend_train_rows(n, nv)
    r_a = [torch.zeros(rows_a, ev(k + 1), device=dev) for k in ins]
    l_a = [torch.zeros(rows_a, ev(m), device=dev) for m in outs]
    gf_a, sp_a = torch.zeros(n, s, f, device=dev), torch.zeros(rows_a // 32, device=dev)
    L.call("gens_blend_train_bwd", *args, L.ptr_table(r_a), L.ptr_table(l_a), L.ptr(gf_a), L.ptr(sp_a), L.stream())
    # transposed, with its operand rows dumped
    g_lanes = 2 if nv == 3 else 4
    rows_b = 16 * (-(-n // (16 // g_lanes)))
    r_b = [torch.zeros(rows_b, ev(k + 1), device=dev) for k in ins]
    l_b = [torch.zeros(rows_b, ev(m), device=dev) for m in outs]
    csz, n_parts = lib.gens_blend_train_acc_floats(n_levels), lib.gens_blend_train_t_parts(n, nv)
    assert n_parts > 0
    gf_b, sp_b = torch.zeros(n, s, f, device=dev), torch.zeros(n_parts, device=dev)
    parts, cc = torch.zeros(n_parts, csz, device=dev), torch.zeros(csz, device=dev)
    L.call("gens_blend_train_bwd_t_dump", *args, L.ptr(gf_b), L.ptr(sp_b), L.ptr(parts), L.ptr(cc), L.ptr_table(r_b), L.ptr_table(l_b), L.stream())
    # the same launch without the dump: the same sums
    gf_c, sp_c = torch.zeros(n, s, f, device=dev), torch.zeros(n_parts, device=dev)
    parts_c, cc_c = torch.zeros(n_parts, csz, device=dev), torch.zeros(csz, device=dev)
    L.call("gens_blend_train_bwd_t", *args, L.ptr(gf_c), L.ptr(sp_c), L.ptr(parts_c), L.ptr(cc_c), L.stream())
    torch.cuda.synchronize()
    # (point, view) -> its row in either kernel
    pt = torch.arange(n_live, device=dev).repeat_interleave(s)
    vw = torch.arange(s, device=dev).repeat(n_live)
    ppw_a = 32 // s
    row_a = (pt // ppw_a) * 32 + (pt % ppw_a) * s + vw
    ppw_b = 16 // g_lanes
    row_b = (pt // ppw_b) * 16 + (pt % ppw_b) * g_lanes + vw
    names = ["ray_dir_fc.0", "ray_dir_fc.2", "base_fc.0", "base_fc.2", "vis_fc.0", "vis_fc.2", "vis_fc2.0", "vis_fc2.2", "rgb_fc.0", "rgb_fc.2", "rgb_fc.4"]
    worst = {}
    for l in range(11):
        for kind, a, b in (("R", r_a[l][row_a], r_b[l][row_b]), ("L", l_a[l][row_a], l_b[l][row_b])):
            scale = max(float(a.abs().max()), 1e-6)
            err = float((a - b).abs().max()) / scale
            worst[f"{names[l]}.{kind}"] = err
    print({k: f"{v:.1e}" for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if not v <= 3e-5}
    assert not bad, bad
    assert float((gf_a[:n_live] - gf_b[:n_live]).abs().max()) <= 3e-5 * max(float(gf_a[:n_live].abs().max()), 1e-6)
    s_a, s_b = float(sp_a.sum()), float(sp_b.sum())
    top_l = max(float(t[row_a].abs().max()) for t in l_a)
    # d loss / d |s|: per point a difference of near-equal terms divided by the sum of the raw view weights -- with TWO source views that sum is ONE
    # difference of two exponentials, ~1e-6 where the viewing angles agree, and float32 round-off in w_bar comes out multiplied by 1e6 (the oracle
    # test's remark on two views): held to the scale of the cotangents, loosely for two views
    assert abs(s_a - s_b) <= (1e-3 if nv == 3 else 1e-4) * max(abs(s_a), top_l), (s_a, s_b)
    assert torch.equal(cc, cc_c) and torch.equal(sp_b, sp_c) and torch.equal(gf_b, gf_c)             # deterministic, dump or not
    # the blocks: [dW_l | db_l] = L^T [R | 1] over the live rows (float64 on the row-major kernel's operand rows)
    off = 0
    for l in range(11):
        m, k = ev(outs[l]), ev(ins[l] + 1)
        ref = (l_a[l][row_a].double().T @ r_a[l][row_a].double())[:outs[l], :ins[l] + 1]
        got = cc[off:off + m * k].view(m, k)[:outs[l], :ins[l] + 1].double()
        err = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-9)
        assert err <= 2e-5, (names[l], err)
        off += m * k
    assert off == csz
