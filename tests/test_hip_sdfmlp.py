"""GPU parity of the fused SDF-network kernel (gens_sdf_mlp) against the PyTorch layers on the K2 look-up kernels, and
against the CPU oracle's functional MLP."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(n_levels, seed):
    from gens_amd.config import gens_model_conf
    from gens_amd.models.modules.sdf_network import SDFNetwork
    torch.manual_seed(seed)
    dims = (16, 12, 8, 6, 4)[:n_levels]
    net = SDFNetwork(**gens_model_conf(volume_dims=dims)["implicit_surface"]["sdf_network"])
    with torch.no_grad():
        for p in net.parameters():                       # make the conditioning channels and every bias matter
            p.add_(0.05 * torch.randn_like(p) * (p.abs().mean() + 0.02))
    return net.cuda(), dims


@pytest.mark.parametrize("n_levels,n", [(3, 1000), (5, 777), (3, 31), (3, 1), (1, 500), (2, 333), (4, 650), (1, 1), (4, 33)])
def test_fused_sdf_and_gradient_match_torch_layers(n_levels, n):
    from gens_amd import ops, synthetic
    net, dims = _net(n_levels, seed=n_levels)
    vols = [v.cuda() * 3 for v in synthetic.make_volumes(dims, seed=9)]
    g = torch.Generator().manual_seed(n)
    pts = (torch.rand(n, 3, generator=g) * 2.2 - 1.1).cuda()          # some points outside the cube (zero padding)
    packed = ops.VolumeSet.packed(vols)
    x = pts.clone().requires_grad_(True)
    ref = net.sdf(x, packed)
    ref_g = torch.autograd.grad(ref, x, torch.ones_like(ref))[0]
    plan = ops.SdfMlpPlan(net)
    sdf, grad = ops.sdf_mlp(plan, packed, pts, want_grad=True)
    only = ops.sdf_mlp(plan, packed, pts)
    assert (sdf - ref).abs().max() < 2e-5, (sdf - ref).abs().max()
    assert (only - ref).abs().max() < 2e-5
    assert (grad - ref_g).abs().max() < 2e-4 * max(1.0, ref_g.abs().max().item()), (grad - ref_g).abs().max()


def test_fused_indexed_scatter_matches_masked_evaluation():
    """The masked evaluation of implicit_surface.py:175-191: only selected points are evaluated, the rest keep 100 / 0."""
    from gens_amd import ops, synthetic
    net, dims = _net(3, seed=5)
    vols = [v.cuda() for v in synthetic.make_volumes(dims, seed=2)]
    packed = ops.VolumeSet.packed(vols)
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(500, 3, generator=g) * 2 - 1).cuda()
    idx = torch.nonzero((torch.rand(500, generator=g) > 0.4).cuda())[:, 0]
    plan = ops.SdfMlpPlan(net)
    sdf = torch.full((500, 1), 100.0, device="cuda")
    grad = torch.zeros(500, 3, device="cuda")
    ops.sdf_mlp(plan, packed, pts, index=idx, want_grad=True, sdf_out=sdf, grad_out=grad)
    dense_s, dense_g = ops.sdf_mlp(plan, packed, pts, want_grad=True)
    keep = torch.zeros(500, dtype=torch.bool, device="cuda")
    keep[idx] = True
    assert torch.equal(sdf[keep], dense_s[keep]) and torch.equal(grad[keep], dense_g[keep])
    assert (sdf[~keep] == 100).all() and (grad[~keep] == 0).all()


def test_fused_matches_cpu_oracle(golden):
    from gens_amd import ops
    from oracle import render_oracle as R
    from tests.test_hip_render import build_surface
    g = golden("g9b_render")
    surf = build_surface(g)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    vols = [g[f"vol{i}"] for i in range(3)]
    pts = torch.rand(300, 3, generator=torch.Generator().manual_seed(1)) * 1.8 - 0.9
    ref = R.sdf_mlp(sd, pts, vols)[:, :1]
    ref_g, _ = R.sdf_gradient(sd, pts, vols, second=False)
    plan = ops.SdfMlpPlan(surf.sdf_network)
    s, gr = ops.sdf_mlp(plan, ops.VolumeSet.packed([v.cuda() for v in vols]), pts.cuda(), want_grad=True)
    assert (s.cpu() - ref).abs().max() < 2e-5
    assert (gr.cpu() - ref_g.detach()).abs().max() < 2e-4


@pytest.mark.parametrize("n_levels,n", [(3, 1000), (5, 777), (3, 33), (3, 1), (3, 129)])
def test_split_half_kernel_matches_float32_kernel(n_levels, n):
    """gens_sdf_value_f16 and gens_sdf_grad_f16: (hi, lo) half operands, three f16 MFMAs per product, float32 accumulation."""
    from gens_amd import ops, synthetic
    net, dims = _net(n_levels, seed=10 + n_levels)
    vols = ops.VolumeSet.packed([v.cuda() * 3 for v in synthetic.make_volumes(dims, seed=9)])
    pts = (torch.rand(n, 3, generator=torch.Generator().manual_seed(n)) * 2.2 - 1.1).cuda()
    plan = ops.SdfMlpPlan(net)
    s32, g32 = ops.sdf_mlp(plan, vols, pts, want_grad=True)
    s16, g16 = ops.sdf_mlp(plan, vols, pts, want_grad=True, precision="f16x2")
    only = ops.sdf_mlp(plan, vols, pts, precision="f16x2")
    assert not plan.overflowed()
    assert (only - s32).abs().max() < 1e-5
    assert plan.grad_pieces is not None and not torch.equal(g16, g32)           # (the split-half kernel ran)
    assert (s16 - s32).abs().max() < 1e-5                                       # measured 1.1e-6
    assert (g16 - g32).abs().max() < 2e-5 * max(1.0, float(g32.abs().max()))    # measured 2.0e-6 at |grad| <= 1.15


def test_split_half_gradient_kernel_scatters_like_the_float32_kernel(monkeypatch):
    """Index map + device-side count: the rows the launch does not own keep their defaults, the owned ones match the float32 kernel; and
    kernels.sdf_grad_f16 = False keeps the value + gradient pass on the float32 kernel bit for bit."""
    from gens_amd import ops, synthetic
    net, dims = _net(3, seed=21)
    vols = ops.VolumeSet.packed([v.cuda() for v in synthetic.make_volumes(dims, seed=3)])
    g = torch.Generator().manual_seed(4)
    pts = (torch.rand(700, 3, generator=g) * 2.2 - 1.1).cuda()
    idx = torch.randperm(700, generator=g)[:400].cuda()
    count = torch.tensor([333], dtype=torch.int32, device="cuda")
    plan = ops.SdfMlpPlan(net)
    outs = {}
    for prec in ("f32", "f16x2"):
        sdf, grad = torch.full((700, 1), 100.0, device="cuda"), torch.full((700, 3), -7.0, device="cuda")
        ops.sdf_mlp(plan, vols, pts, index=idx, want_grad=True, sdf_out=sdf, grad_out=grad, precision=prec, count=count)
        outs[prec] = (sdf, grad)
    assert not plan.overflowed()
    owned = torch.zeros(700, dtype=torch.bool, device="cuda")
    owned[idx[:333]] = True
    assert torch.equal(outs["f16x2"][0][~owned], outs["f32"][0][~owned]) and bool((outs["f16x2"][1][~owned] == -7.0).all())
    assert (outs["f16x2"][0][owned] - outs["f32"][0][owned]).abs().max() < 1e-5
    assert (outs["f16x2"][1][owned] - outs["f32"][1][owned]).abs().max() < 2e-5 * max(1.0, float(outs["f32"][1].abs().max()))
    assert not torch.equal(outs["f16x2"][1][owned], outs["f32"][1][owned])       # (it IS the other kernel)
    monkeypatch.setattr(ops.kernels, "sdf_grad_f16", False)
    s, gr = ops.sdf_mlp(plan, vols, pts, want_grad=True, precision="f16x2")
    s32, g32 = ops.sdf_mlp(plan, vols, pts, want_grad=True)
    assert torch.equal(s, s32) and torch.equal(gr, g32)


def test_split_half_gradient_kernel_many_workgroups_reuse_their_slots():
    """More workgroups than CUs (every (CU, wave) slot of the stash is taken and released many times), a ragged tail, twice in a row on
    the same stash: the same answer as the float32 kernel everywhere."""
    from gens_amd import ops, synthetic
    net, dims = _net(3, seed=5)
    vols = ops.VolumeSet.packed([v.cuda() for v in synthetic.make_volumes(dims, seed=2)])
    n = 128 * 256 * 5 + 77
    pts = (torch.rand(n, 3, generator=torch.Generator().manual_seed(8)) * 2.2 - 1.1).cuda()
    plan = ops.SdfMlpPlan(net)
    s32, g32 = ops.sdf_mlp(plan, vols, pts, want_grad=True)
    for _ in range(2):
        s16, g16 = ops.sdf_mlp(plan, vols, pts, want_grad=True, precision="f16x2")
        assert (s16 - s32).abs().max() < 1e-5
        assert (g16 - g32).abs().max() < 2e-5 * max(1.0, float(g32.abs().max()))
    assert not plan.overflowed()


def test_split_half_overflow_is_flagged():
    from gens_amd import ops, synthetic
    net, dims = _net(3, seed=1)
    big = [v.cuda() * 1e6 for v in synthetic.make_volumes(dims, seed=9)]     # features far outside the half range
    plan = ops.SdfMlpPlan(net)
    ops.sdf_mlp(plan, ops.VolumeSet.packed(big), torch.zeros(64, 3, device="cuda"), precision="f16x2")
    assert plan.overflowed() and not plan.overflowed()                      # reading the flag clears it
    ops.sdf_mlp(plan, ops.VolumeSet.packed(big), torch.zeros(64, 3, device="cuda"), want_grad=True, precision="f16x2")
    assert plan.overflowed() and not plan.overflowed()                      # the value + gradient kernel raises the same flag
    vols = [v.cuda() for v in synthetic.make_volumes(dims, seed=9)]
    vols[1][0, 2, 3:6, 3:6, 3:6] = float("nan")                             # not-a-number inputs: flagged, the caller re-runs in float32
    pts = (torch.rand(4000, 3, generator=torch.Generator().manual_seed(5)) * 2 - 1).cuda()
    ops.sdf_mlp(plan, ops.VolumeSet.packed(vols), pts, want_grad=True, precision="f16x2")
    assert plan.overflowed()
    plan.grad_scale = 2.0 ** 30                                              # gradients pushed out of the half range: flagged
    ops.sdf_mlp(plan, ops.VolumeSet.packed([v.cuda() for v in synthetic.make_volumes(dims, seed=9)]), pts, want_grad=True, precision="f16x2")
    assert plan.overflowed()


@pytest.mark.parametrize("n_levels,n", [(3, 1), (3, 33), (3, 129), (3, 4097), (5, 257)])
def test_transposed_kernels_equal_row_major_kernel(n_levels, n, monkeypatch):
    """gens_sdf_value / gens_sdf_grad (one wave per 32 points, activations chained in registers) against gens_sdf_mlp (four waves per
    32 points, activations through LDS): the same float32 MFMA products in another summation order, with an index map and a device-side
    count as the masked evaluation of implicit_surface.py:125,179-191 passes them; untouched outputs keep their fill values."""
    from gens_amd import ops, synthetic
    net, dims = _net(n_levels, seed=20 + n_levels)
    packed = ops.VolumeSet.packed([v.cuda() * 2 for v in synthetic.make_volumes(dims, seed=4)])
    g = torch.Generator().manual_seed(n)
    pts = (torch.rand(n, 3, generator=g) * 2.4 - 1.2).cuda()
    idx = torch.randperm(n, generator=g).cuda()
    count = torch.tensor([max(1, (2 * n) // 3)], dtype=torch.int32, device="cuda")
    plan = ops.SdfMlpPlan(net)

    def run():
        sdf, grad, val = torch.full((n, 1), 100.0, device="cuda"), torch.zeros(n, 3, device="cuda"), torch.full((n, 1), 100.0, device="cuda")
        ops.sdf_mlp(plan, packed, pts, index=idx, want_grad=True, sdf_out=sdf, grad_out=grad, count=count)
        ops.sdf_mlp(plan, packed, pts, index=idx, sdf_out=val, count=count)
        return sdf, grad, val

    new = run()
    monkeypatch.setattr(ops.kernels, "sdf_value", "rowmajor")
    monkeypatch.setattr(ops.kernels, "sdf_grad", "rowmajor")
    old = run()
    live = idx[:int(count)]
    dead = idx[int(count):]
    for a, b, tol in zip(new, old, (2e-6, 2e-5, 2e-6)):
        assert (a[live] - b[live]).abs().max() <= tol * max(1.0, b[live].abs().max().item())
        assert torch.equal(a[dead], b[dead])                      # fill values
    assert (new[0][live] - new[2][live]).abs().max() <= 2e-6      # the value of the gradient kernel and of the value kernel


def test_new_sdf_kernels_reject_bad_arguments():
    from gens_amd import lib as L, ops, synthetic
    net, dims = _net(3, seed=1)
    packed = ops.VolumeSet.packed([v.cuda() for v in synthetic.make_volumes(dims, seed=4)])
    plan = ops.SdfMlpPlan(net)
    pts = torch.zeros(8, 3, device="cuda")
    out = torch.zeros(8, 1, device="cuda")
    with pytest.raises(RuntimeError, match="null weight stream"):
        L.call("gens_sdf_value", packed.table, packed.dim_table, 3, None, L.ptr(plan.value_row), 0.0, 1.0, L.ptr(pts), None, 8, None, L.ptr(out),
               L.stream())
    with pytest.raises(RuntimeError, match="scale must be non-zero"):
        L.call("gens_sdf_grad", packed.table, packed.dim_table, 3, L.ptr(plan.grad_stream), L.ptr(plan.grad_row), 0.0, 0.0, L.ptr(pts), None, 8,
               None, L.ptr(out), L.ptr(pts), L.ptr(ops.sdf_grad_stash("cuda"), torch.uint8), L.stream())
    six = ops.VolumeSet.packed([v.cuda() for v in synthetic.make_volumes((8, 6, 4, 4, 4, 4), seed=4)])
    with pytest.raises(RuntimeError, match="1 to 5 volume levels"):
        L.call("gens_sdf_grad", six.table, six.dim_table, 6, L.ptr(plan.grad_stream), L.ptr(plan.grad_row), 0.0, 1.0, L.ptr(pts), None, 8,
               None, L.ptr(out), L.ptr(pts), L.ptr(ops.sdf_grad_stash("cuda"), torch.uint8), L.stream())
    with pytest.raises(RuntimeError, match="stash"):
        L.call("gens_sdf_grad", packed.table, packed.dim_table, 3, L.ptr(plan.grad_stream), L.ptr(plan.grad_row), 0.0, 1.0, L.ptr(pts), None, 8,
               None, L.ptr(out), L.ptr(pts), None, L.stream())
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    g16 = lambda n_levels=3, pieces=L.ptr(plan.grad_pieces, torch.float16), scale=1.0, g_scale=1.0, stash=L.ptr(ops.sdf_grad_f16_stash("cuda"), torch.uint8), fl=L.ptr(flag, torch.int32): \
        L.call("gens_sdf_grad_f16", packed.table, packed.dim_table, n_levels, pieces, L.ptr(plan.grad_row), 0.0, scale, g_scale, L.ptr(pts), None, 8, None,
               L.ptr(out), L.ptr(pts), stash, fl, L.stream())
    with pytest.raises(RuntimeError, match="built for 3 or 5 volume levels"):
        g16(n_levels=2)
    with pytest.raises(RuntimeError, match="null weight stream / flag"):
        g16(pieces=None)
    with pytest.raises(RuntimeError, match="null weight stream / flag"):
        g16(fl=None)
    with pytest.raises(RuntimeError, match="g_scale positive"):
        g16(g_scale=0.0)
    with pytest.raises(RuntimeError, match="stash"):
        g16(stash=None)
    # empty launches are no-ops
    ops.sdf_mlp(plan, packed, torch.zeros(0, 3, device="cuda"), want_grad=True, precision="f16x2")
    ops.sdf_mlp(plan, packed, torch.zeros(0, 3, device="cuda"), want_grad=True)
    ops.sdf_mlp(plan, packed, torch.zeros(0, 3, device="cuda"))
    ops.sdf_mlp(plan, packed, torch.zeros(0, 3, device="cuda"), precision="f16x2")


def test_transposed_kernels_propagate_not_a_number_inputs():
    """A NaN in a volume texel (or in a point) must come out as NaN, as through the reference's layers: the max / median forms of the
    activation in gens_sdf_value / gens_sdf_grad would otherwise drop it (they carry a poison term for this)."""
    from gens_amd import ops, synthetic
    net, dims = _net(3, seed=4)
    vols = [v.cuda() for v in synthetic.make_volumes(dims, seed=6)]
    vols[1][0, 2, 3:6, 3:6, 3:6] = float("nan")                    # one channel of a block of level-1 texels
    packed = ops.VolumeSet.packed(vols)
    g = torch.Generator().manual_seed(5)
    pts = (torch.rand(4000, 3, generator=g) * 2 - 1).cuda()
    pts[7, 1] = float("nan")
    plan = ops.SdfMlpPlan(net)
    val = ops.sdf_mlp(plan, packed, pts)
    sdf, grad = ops.sdf_mlp(plan, packed, pts, want_grad=True)
    os_env = __import__("os").environ
    os_env["GENS_SDF_VALUE_ROWMAJOR"] = os_env["GENS_SDF_GRAD_ROWMAJOR"] = "1"
    try:
        ref_s, ref_g = ops.sdf_mlp(plan, packed, pts, want_grad=True)
    finally:
        del os_env["GENS_SDF_VALUE_ROWMAJOR"], os_env["GENS_SDF_GRAD_ROWMAJOR"]
    bad = torch.isnan(ref_s[:, 0])
    assert bad[7] and 5 < int(bad.sum()) < 2000
    assert torch.equal(torch.isnan(val[:, 0]), bad) and torch.equal(torch.isnan(sdf[:, 0]), bad)
    assert torch.equal(torch.isnan(grad).any(1), torch.isnan(ref_g).any(1))
    assert (sdf[~bad] - ref_s[~bad]).abs().max() < 2e-6


def test_non_finite_weights_give_not_a_number_outputs():
    from gens_amd import ops, synthetic
    net, dims = _net(3, seed=8)
    with torch.no_grad():
        net.lin4.bias[5] = float("nan")
    packed = ops.VolumeSet.packed([v.cuda() for v in synthetic.make_volumes(dims, seed=6)])
    pts = (torch.rand(100, 3) * 2 - 1).cuda()
    idx = torch.arange(0, 100, 2).cuda()
    count = torch.tensor([30], dtype=torch.int32, device="cuda")
    plan = ops.SdfMlpPlan(net)
    sdf = torch.full((100, 1), 100.0, device="cuda")
    grad = torch.zeros(100, 3, device="cuda")
    ops.sdf_mlp(plan, packed, pts, index=idx, want_grad=True, sdf_out=sdf, grad_out=grad, count=count)
    assert torch.isnan(sdf[idx[:30]]).all() and torch.isnan(grad[idx[:30]]).all()
    assert (sdf[idx[30:]] == 100).all() and (sdf[1::2] == 100).all()
    assert torch.isnan(ops.sdf_mlp(plan, packed, pts)).all()


@pytest.mark.parametrize("n_levels", [1, 2, 4])
def test_other_level_counts_match_the_cpu_oracle(n_levels):
    """Level counts other than the shipped 3 / 5 (BASELINE config[0] is ONE volume): value and gradient of the fused kernels against the oracle's
    functional network (oracle/render_oracle.py::sdf_mlp, sdf_gradient) in float64, points inside and outside the cube, and the value-only kernel
    (gens_sdf_value) bit-identical to the value of the value + gradient kernel (gens_sdf_grad) -- the sampling passes and render_core see ONE SDF."""
    from gens_amd import ops, synthetic
    from oracle import render_oracle as R
    net, dims = _net(n_levels, seed=30 + n_levels)
    vols = synthetic.make_volumes(dims, seed=11)
    sd = {"sdf_network." + k: v.detach().cpu().double() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(n_levels)
    pts = torch.rand(700, 3, generator=g) * 2.3 - 1.15
    vols64 = [v.double() * 2 for v in vols]
    ref = R.sdf_mlp(sd, pts.double(), vols64)[:, :1]
    ref_g, _ = R.sdf_gradient(sd, pts.double(), vols64, second=False)
    plan = ops.SdfMlpPlan(net)
    assert plan.n_levels == n_levels and plan.value_units is None and plan.grad_pieces is None      # (split-half kernels: 3 and 5 levels only)
    packed = ops.VolumeSet.packed([(v * 2).cuda() for v in vols])
    s, gr = ops.sdf_mlp(plan, packed, pts.cuda(), want_grad=True)
    only = ops.sdf_mlp(plan, packed, pts.cuda())
    assert float((s.cpu().double() - ref).abs().max()) < 2e-5
    assert float((gr.cpu().double() - ref_g.detach()).abs().max()) < 2e-4 * max(1.0, float(ref_g.abs().max()))
    assert torch.equal(only, s)
    # the opt-in split-half precision falls back to float32 for these level counts: same numbers, no error
    s16, g16 = ops.sdf_mlp(plan, packed, pts.cuda(), want_grad=True, precision="f16x2")
    assert torch.equal(s16, s) and torch.equal(g16, gr)
    assert torch.equal(ops.sdf_mlp(plan, packed, pts.cuda(), precision="f16x2"), only)
