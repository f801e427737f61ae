"""The reference's sampler boundary in full generality (cuda_gridsample.py:7-14: grid_sample_2d / grid_sample_3d, 'zeros' | 'border', either
align_corners; gridsample_cuda.cpp:26-56: grad2_2d / grad2_3d).

CPU: oracle/grid_sample_oracle.py against ATen itself -- F.grid_sample and its autograd for value and first derivatives in every mode, float64
central differences of ATen's backward for the second derivatives (the reference's CUDA second-order kernels cannot be built here).
GPU: the general kernels K20 (gens_grid_sample_*) and the compat module's Function pair against that oracle."""
import itertools

import pytest
import torch
import torch.nn.functional as F

from oracle import grid_sample_oracle as G

MODES = list(itertools.product((2, 3), ("zeros", "border"), (True, False)))


def _case(dim, seed, dtype=torch.float32, n=2, c=3, n_pts=160):
    g = torch.Generator().manual_seed(seed)
    spatial = (5, 7) if dim == 2 else (4, 5, 6)
    x = torch.randn(n, c, *spatial, generator=g, dtype=torch.float64).to(dtype)
    # points inside, on the faces, outside (up to 1.4) and exactly on source-index integers
    grid = torch.rand(n, n_pts, dim, generator=g, dtype=torch.float64) * 2.8 - 1.4
    grid[:, :8] = torch.tensor([-1.0, 1.0, 0.0, -1.0, 1.0, 0.5, -0.5, 1.0]).reshape(1, 8, 1).expand(n, 8, dim) * torch.tensor([1.0, -1.0, 1.0][:dim])
    grid = grid.reshape(n, *((8, n_pts // 8) if dim == 2 else (4, 2, n_pts // 8)), dim).to(dtype)
    go = torch.randn(n, c, *grid.shape[1:-1], generator=g, dtype=torch.float64).to(dtype)
    ggi = torch.randn(n, c, *spatial, generator=g, dtype=torch.float64).to(dtype)
    ggg = torch.randn(*grid.shape, generator=g, dtype=torch.float64).to(dtype)
    return x, grid, go, ggi, ggg


def _oracle_all_orders(x, grid, go, ggi, ggg, pad, ac):
    """-> out, (gI, gG), (ggO, gI2, gG2) of oracle.sample by autograd, with the cotangents the reference's Function pair would see."""
    x, grid, go = x.clone().requires_grad_(True), grid.clone().requires_grad_(True), go.clone().requires_grad_(True)
    out = G.sample(x, grid, pad, ac)
    g_in, g_grid = torch.autograd.grad(out, [x, grid], go, create_graph=True)
    s = (g_in * ggi).sum() + (g_grid * ggg).sum()
    second = torch.autograd.grad(s, [go, x, grid])
    return out.detach(), (g_in.detach(), g_grid.detach()), tuple(t.detach() for t in second)


@pytest.mark.parametrize("dim,pad,ac", MODES)
def test_oracle_sampler_is_atens_sampler_to_first_order(dim, pad, ac):
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
        x, grid, go, _, _ = _case(dim, 10 * dim + ac, dtype)
        x.requires_grad_(True)
        grid.requires_grad_(True)
        a, b = G.sample(x, grid, pad, ac), F.grid_sample(x, grid, mode="bilinear", padding_mode=pad, align_corners=ac)
        assert float((a - b).abs().max()) <= tol
        ga, gb = torch.autograd.grad(a, [x, grid], go), torch.autograd.grad(b, [x, grid], go)
        assert float((ga[0] - gb[0]).abs().max()) <= tol * 10 and float((ga[1] - gb[1]).abs().max()) <= tol * 50


@pytest.mark.parametrize("dim,pad,ac", MODES)
def test_oracle_second_order_is_the_derivative_of_atens_backward(dim, pad, ac):
    """ggO, gI', gG' of the oracle against float64 central differences of ATen's OWN first backward, at points away from the cell faces
    (where the interpolant has a kink and a finite difference straddles it)."""
    x, grid, go, ggi, ggg = _case(dim, 20 * dim + ac, torch.float64, n=1, c=2, n_pts=64)
    spatial = x.shape[2:]
    keep = torch.ones(grid.shape[:-1], dtype=torch.bool)
    for a in range(dim):
        s = G.source_index(grid[..., a], spatial[dim - 1 - a], "zeros", ac)
        keep &= ((s - torch.round(s)).abs() > 1e-3)
    _, _, (gg_out, g_in2, g_grid2) = _oracle_all_orders(x, grid, go, ggi, ggg, pad, ac)

    def first_backward(x_, grid_, go_):
        x_, grid_ = x_.clone().requires_grad_(True), grid_.clone().requires_grad_(True)
        out = F.grid_sample(x_, grid_, mode="bilinear", padding_mode=pad, align_corners=ac)
        g_in, g_grid = torch.autograd.grad(out, [x_, grid_], go_)
        return float((g_in * ggi).sum() + (g_grid * ggg).sum())
    h = 1e-6
    # d/d grid: perturb a few entries
    flat = grid.reshape(-1)
    for j in torch.nonzero(keep.reshape(-1))[:12, 0].tolist():
        for a in range(dim):
            e = torch.zeros_like(flat)
            e[j * dim + a] = h
            fd = (first_backward(x, (flat + e).reshape(grid.shape), go) - first_backward(x, (flat - e).reshape(grid.shape), go)) / (2 * h)
            assert abs(fd - float(g_grid2.reshape(-1)[j * dim + a])) <= 1e-6 * max(1.0, abs(fd)), (j, a, fd, float(g_grid2.reshape(-1)[j * dim + a]))
    # d/d grad_output and d/d input: the first backward is LINEAR in both, one difference each is exact
    for t, got in ((go, gg_out), (x, g_in2)):
        for j in range(0, t.numel(), max(1, t.numel() // 10)):
            e = torch.zeros_like(t).reshape(-1)
            e[j] = 1.0
            e = e.reshape(t.shape)
            fd = (first_backward(x + e if t is x else x, grid, go + e if t is go else go) - first_backward(x - e if t is x else x, grid, go - e if t is go else go)) / 2
            assert abs(fd - float(got.reshape(-1)[j])) <= 1e-9 * max(1.0, abs(fd)), (j, fd, float(got.reshape(-1)[j]))


# ---------------------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------------------
def _close(a, b, tol, what):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    assert float((a - b).abs().max()) <= tol * scale, f"{what}: max err {float((a - b).abs().max()):.3e} (scale {scale:.2e})"


@pytest.mark.gpu
@pytest.mark.parametrize("dim,pad,ac", MODES)
def test_general_sampler_function_pair_matches_the_oracle_to_second_order(dim, pad, ac):
    """grid_sample_2d / grid_sample_3d of the compat module exactly as a caller of the reference's file would use them: value, first
    derivatives (autograd through the Function pair) and second derivatives (the grad2 kernels), against the float64 oracle."""
    from gens_amd.compat import cuda_gridsample as cug
    x, grid, go, ggi, ggg = _case(dim, 30 * dim + ac)
    ref_out, (ref_gi, ref_gg), (ref_ggo, ref_gi2, ref_gg2) = _oracle_all_orders(x.double(), grid.double(), go.double(), ggi.double(), ggg.double(), pad, ac)
    fn = cug.grid_sample_2d if dim == 2 else cug.grid_sample_3d
    xd, gd, god = x.cuda().requires_grad_(True), grid.cuda().requires_grad_(True), go.cuda().requires_grad_(True)
    out = fn(xd, gd, padding_mode=pad, align_corners=ac)
    _close(out, ref_out, 1e-5, "value")
    g_in, g_grid = torch.autograd.grad(out, [xd, gd], god, create_graph=True)
    _close(g_in, ref_gi, 2e-5, "grad_input")
    _close(g_grid, ref_gg, 1e-4, "grad_grid")
    s = (g_in * ggi.cuda()).sum() + (g_grid * ggg.cuda()).sum()
    gg_out, g_in2, g_grid2 = torch.autograd.grad(s, [god, xd, gd])
    _close(gg_out, ref_ggo, 1e-4, "grad_grad_output")
    _close(g_in2, ref_gi2, 1e-4, "second grad_input")
    _close(g_grid2, ref_gg2, 5e-4, "second grad_grid")
    # the extension's entry point itself, in its tensor layouts, with and without grad2_grad_input (cuda_gridsample.py:113-114)
    entry = cug.grad2_2d if dim == 2 else cug.grad2_3d
    outs = entry(ggi.cuda(), ggg.cuda(), go.cuda(), x.cuda(), grid.cuda(), ["zeros", "border"].index(pad), ac)
    _close(outs[0], ref_ggo, 1e-4, "grad2: grad_grad_output")
    _close(outs[1], ref_gi2, 1e-4, "grad2: grad_input")
    _close(outs[2], ref_gg2, 5e-4, "grad2: grad_grid")
    _, _, (ref_ggo0, ref_gi20, ref_gg20) = _oracle_all_orders(x.double(), grid.double(), go.double(), torch.zeros_like(ggi).double(), ggg.double(), pad, ac)
    outs = entry(None, ggg.cuda(), go.cuda(), x.cuda(), grid.cuda(), ["zeros", "border"].index(pad), ac)
    _close(outs[0], ref_ggo0, 1e-4, "grad2 without grad2_grad_input: grad_grad_output")
    _close(outs[2], ref_gg20, 5e-4, "grad2 without grad2_grad_input: grad_grid")


@pytest.mark.gpu
def test_general_sampler_agrees_with_the_fast_lookup_on_the_hot_paths_call():
    """The call lookup_volume makes (batch 1, C = 4, zeros, align_corners=True, 3-D) is served by K2; forced through K20 it gives the same
    numbers (two kernels, one function), and so do edge inputs: an empty grid, infinite and NaN coordinates."""
    from gens_amd.compat import cuda_gridsample as cug
    g = torch.Generator().manual_seed(5)
    vol = torch.randn(1, 4, 9, 8, 7, generator=g).cuda()
    grid = (torch.rand(1, 1, 1, 500, 3, generator=g) * 2.6 - 1.3).cuda()
    fast = cug.grid_sample_3d(vol, grid)
    general = cug._GridSampleForward.apply(vol, grid, 0, True)
    _close(fast, general, 2e-6, "K2 vs K20")
    assert cug.grid_sample_2d(torch.randn(2, 3, 4, 5).cuda(), torch.zeros(2, 0, 7, 2).cuda()).shape == (2, 3, 0, 7)
    weird = torch.tensor([[float("inf"), 0.0], [float("nan"), 0.1], [-float("inf"), 2.0], [0.3, float("nan")]]).reshape(1, 1, 4, 2).cuda()
    img = torch.randn(1, 2, 4, 5).cuda()
    for pad in ("zeros", "border"):
        got = cug.grid_sample_2d(img, weird, padding_mode=pad)
        assert torch.isfinite(got).all()
    # finite but absurdly far points: ATen skips out-of-bounds taps (zeros: exactly 0; border: the clamped texel) -- no 0 * inf from weights
    far2 = torch.tensor([[1e30, 0.0], [-3e38, 1e20], [0.2, -1e15], [1e12, 1e12]]).reshape(1, 1, 4, 2).cuda()
    far3 = torch.tensor([[1e30, 0.0, 0.1], [-3e38, 1e20, 0.0], [0.2, -1e15, 0.3], [1e12, 1e12, 1e12]]).reshape(1, 1, 1, 4, 3).cuda()
    for pad in ("zeros", "border"):
        for ac in (True, False):
            _close(cug.grid_sample_2d(img, far2, padding_mode=pad, align_corners=ac), F.grid_sample(img, far2, padding_mode=pad, align_corners=ac), 1e-6, f"far 2-D {pad} {ac}")
            _close(cug.grid_sample_3d(vol, far3, padding_mode=pad, align_corners=ac), F.grid_sample(vol, far3, padding_mode=pad, align_corners=ac), 1e-6, f"far 3-D {pad} {ac}")
    with pytest.raises(RuntimeError, match="padding_mode"):
        cug.grid_sample_2d(img, weird, padding_mode="reflection")
    with pytest.raises(RuntimeError, match="device"):
        cug.grid_sample_2d(img.cpu(), weird.cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("dim,pad,ac", [(2, "border", True), (3, "zeros", True), (3, "border", False)])
def test_reference_function_pair_on_the_ctypes_extension(dim, pad, ac):
    """The route INTEGRATION.md section 3 describes for a host that keeps the reference's OWN cuda_gridsample.py: its Function pair -- forward =
    F.grid_sample, backward = aten::grid_sampler_{2,3}d_backward, backward of the backward = gridsample_grad2.grad2_{2,3}d -- with only the
    extension replaced by gens_amd/compat/gridsample_grad2.py (ctypes on libgens_hip.so).  The pair below follows cuda_gridsample.py:21-123."""
    from gens_amd.compat import gridsample_grad2

    class Backward(torch.autograd.Function):                                       # _GridSample{2,3}dBackward (:45-66, :94-123)
        @staticmethod
        def forward(ctx, grad_output, input, grid, padding_mode, align_corners):
            op = torch.ops.aten.grid_sampler_2d_backward if dim == 2 else torch.ops.aten.grid_sampler_3d_backward
            grad_input, grad_grid = op(grad_output, input, grid, 0, padding_mode, align_corners, (ctx.needs_input_grad[1], ctx.needs_input_grad[2]))
            ctx.save_for_backward(grad_output, input, grid)
            ctx.padding_mode, ctx.align_corners = padding_mode, align_corners
            return grad_input, grad_grid

        @staticmethod
        def backward(ctx, grad2_grad_input, grad2_grad_grid):
            grad_output, input, grid = ctx.saved_tensors
            fn = gridsample_grad2.grad2_2d if dim == 2 else gridsample_grad2.grad2_3d
            out = fn(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, ctx.padding_mode, ctx.align_corners)
            return out[0], out[1], out[2], None, None

    class Forward(torch.autograd.Function):                                        # _GridSample{2,3}dForward (:21-43, :71-91)
        @staticmethod
        def forward(ctx, input, grid, padding_mode, align_corners):
            output = F.grid_sample(input=input, grid=grid, mode="bilinear", padding_mode=padding_mode, align_corners=align_corners)
            ctx.save_for_backward(input, grid)
            ctx.padding_mode, ctx.align_corners = ["zeros", "border"].index(padding_mode), align_corners
            return output

        @staticmethod
        def backward(ctx, grad_output):
            input, grid = ctx.saved_tensors
            grad_input, grad_grid = Backward.apply(grad_output, input, grid, ctx.padding_mode, ctx.align_corners)
            return grad_input, grad_grid, None, None

    x, grid, go, ggi, ggg = _case(dim, 40 * dim + ac)
    _, _, (ref_ggo, ref_gi2, ref_gg2) = _oracle_all_orders(x.double(), grid.double(), go.double(), ggi.double(), ggg.double(), pad, ac)
    xd, gd, god = x.cuda().requires_grad_(True), grid.cuda().requires_grad_(True), go.cuda().requires_grad_(True)
    out = Forward.apply(xd, gd, pad, ac)
    g_in, g_grid = torch.autograd.grad(out, [xd, gd], god, create_graph=True)
    s = (g_in * ggi.cuda()).sum() + (g_grid * ggg.cuda()).sum()
    gg_out, g_in2, g_grid2 = torch.autograd.grad(s, [god, xd, gd])
    _close(gg_out, ref_ggo, 1e-4, "grad_grad_output")
    _close(g_in2, ref_gi2, 1e-4, "second grad_input")
    _close(g_grid2, ref_gg2, 5e-4, "second grad_grid")
