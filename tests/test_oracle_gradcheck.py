"""SURVEY.md section 4, "Gradient tests": finite-difference checks in float64 of the CPU restatement itself (the oracle the
HIP backward / backward-of-backward kernels are compared with), incl. points on the border cells and outside the cube where
the zero padding matters (section 8c option ii)."""
import torch

from oracle import gens_oracle as K

torch.manual_seed(0)


def _pts(n, lo, hi, dims):
    """Random points in [lo, hi]^3 kept away from cell faces (the trilinear interpolant is not differentiable on them)."""
    g = torch.Generator().manual_seed(n)
    p = torch.rand(n, 3, generator=g, dtype=torch.float64) * (hi - lo) + lo
    for d in dims:
        pos = (p + 1) * 0.5 * (d - 1)
        frac = pos - torch.floor(pos)
        p = p + torch.where((frac < 0.05) | (frac > 0.95), 0.2 / (d - 1), 0.0)
    return p


def test_lookup_volume_first_and_second_order_fd64():
    dims = (4, 3)
    vols = [torch.randn(1, 2, d, d, d, dtype=torch.float64, requires_grad=True) for d in dims]
    for lo, hi in ((-0.9, 0.9), (-1.25, 1.25)):            # interior, then border cells + outside (zero padding)
        pts = _pts(5, lo, hi, dims).requires_grad_(True)
        assert torch.autograd.gradcheck(lambda p, *v: K.lookup_volume(list(v), p), (pts, *vols), eps=1e-6, atol=1e-6)
        assert torch.autograd.gradgradcheck(lambda p, *v: K.lookup_volume(list(v), p), (pts, *vols), eps=1e-6, atol=1e-6)


def test_explicit_backward_functions_agree_with_autograd_fd64():
    """lookup_volume_bwd / lookup_volume_bwd2 (what the HIP kernels K2 bwd and K2'' are tested against) vs central differences."""
    dims = (5,)
    vols = [torch.randn(1, 4, 5, 5, 5, dtype=torch.float64)]
    pts = _pts(7, -1.2, 1.2, dims)
    g_out = torch.randn(7, 4, dtype=torch.float64)
    gv, gp = K.lookup_volume_bwd(g_out, vols, pts)
    eps = 1e-6
    for i in range(3):                                      # d <f, g_out> / d pts by central differences
        e = torch.zeros_like(pts)
        e[:, i] = eps
        fd = ((K.lookup_volume(vols, pts + e) - K.lookup_volume(vols, pts - e)) * g_out).sum(-1) / (2 * eps)
        assert torch.allclose(fd, gp[:, i], atol=1e-7)
    gg_pts = torch.randn(7, 3, dtype=torch.float64)
    gg_vols = [torch.randn_like(v) for v in vols]
    ggo, gv2, gp2 = K.lookup_volume_bwd2(gg_vols, gg_pts, g_out, vols, pts)

    def phi(p, v, go):                                      # <gP, gg_pts> + <gV, gg_vols> as a function of the first backward's inputs
        gv_, gp_ = K.lookup_volume_bwd(go, [v], p)
        return (gp_ * gg_pts).sum() + (gv_[0] * gg_vols[0]).sum()
    for i in range(3):
        e = torch.zeros_like(pts)
        e[2, i] = eps
        fd = (phi(pts + e, vols[0], g_out) - phi(pts - e, vols[0], g_out)) / (2 * eps)
        assert abs(float(fd) - float(gp2[2, i])) < 1e-6
    e = torch.zeros_like(g_out)
    e[3, 1] = eps
    fd = (phi(pts, vols[0], g_out + e) - phi(pts, vols[0], g_out - e)) / (2 * eps)
    assert abs(float(fd) - float(ggo[3, 1])) < 1e-6


def test_volume_build_mean_is_linear_in_the_features():
    """K1's mean channels are linear in the features, so a finite difference of ANY step equals the gradient (float32 oracle)."""
    from gens_amd import synthetic
    sc = synthetic.make_scene(nv=3, h=24, w=32, n_levels=1, seed=2)
    feats = [sc["features"][0].clone().requires_grad_(True)]
    vols, _ = K.volume_build(feats, sc["intrs"], sc["c2ws"], [6])
    cot = torch.randn_like(vols[0][:, :4])
    g, = torch.autograd.grad((vols[0][:, :4] * cot).sum(), feats)
    d = torch.randn_like(feats[0])
    v2, _ = K.volume_build([feats[0].detach() + 0.5 * d], sc["intrs"], sc["c2ws"], [6])
    v1, _ = K.volume_build([feats[0].detach() - 0.5 * d], sc["intrs"], sc["c2ws"], [6])
    fd, an = float(((v2[0][:, :4] - v1[0][:, :4]) * cot).sum()), float((g * d).sum())
    assert abs(fd - an) < 1e-3 * max(1.0, abs(an))


def test_second_order_lookup_outside_the_cube_vs_fd64_of_aten():
    """The oracle's K2'' restatement at and beyond the volume border against float64 central differences of ATen's own
    grid_sampler_3d backward (the op the reference calls, cuda_gridsample.py:97): the same case the HIP kernel is held to in
    tests/test_hip_kernels.py::test_k2_second_order_outside_the_cube_vs_fd64_of_aten."""
    from .test_hip_kernels import second_order_fd64_case
    vols, pts, go, ggp, gp2, ggo_ref, gv_ref = second_order_fd64_case()
    ggo, gv2, gp2_o = K.lookup_volume_bwd2(None, ggp.float(), go.float(), [v.float() for v in vols], pts.float())
    assert (gp2_o - gp2).abs().max() < 2e-3 and (ggo - ggo_ref).abs().max() < 1e-4
    for a, b in zip(gv2, gv_ref):
        assert (a - b).abs().max() < 1e-4
