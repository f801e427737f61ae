"""Drop-in for the reference's only native boundary, `models/modules/grid_sample_cuda/cuda_gridsample.py` (+ the `gridsample_grad2`
extension it JIT-builds from gridsample_cuda.cpp / gridsample_cuda.cu), in the REFERENCE'S OWN calling convention -- everything that file
exports:

    grid_sample_2d(input (N,C,H,W),   grid (N,Ho,Wo,2),    padding_mode='zeros'|'border', align_corners=True) -> (N,C,Ho,Wo)
    grid_sample_3d(input (N,C,D,H,W), grid (N,Do,Ho,Wo,3), padding_mode='zeros'|'border', align_corners=True) -> (N,C,Do,Ho,Wo)
        twice differentiable like the reference's Function pairs (cuda_gridsample.py:7-14, 21-66, 71-123): the value is what
        F.grid_sample returns, the first backward what aten::grid_sampler_{2,3}d_backward returns, the second backward what
        grad2_2d / grad2_3d return; the outputs of the second backward are constants, as in the reference.
    grad2_2d / grad2_3d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode, align_corners) -> [ggO, gI, gG]
        the extension's entry points (gridsample_cuda.cpp:26-56).

A maintainer who wants ONLY the sampler replaced (and keeps the reference's projector / networks) puts this file in place of
cuda_gridsample.py.

Two kernel families sit behind it.  The call the reference's hot path makes -- lookup_volume's, projector.py:223-229: batch 1, four-channel
levels, zeros padding, align_corners=True, 3-D -- runs on K2 (gens_lookup_volume_*, csrc/k2_lookup.hip).  Every other combination the
reference's file accepts (2-D, 'border', align_corners=False, batches, other channel counts) runs on the general kernels K20
(gens_grid_sample_*, csrc/k20_grid_sample.hip).  Device float32 tensors only; anything else raises (there is no CPU path).

The grid follows F.grid_sample: last axis (x, y, z) indexes (W, H, D), i.e. input element [d(z)][h(y)][w(x)]; K2 reads element
[ix(px)][iy(py)][iz(pz)] of an (X, Y, Z) volume at point p, so a grid point is handed to it flipped (projector.py:223 flips it the other
way before calling: the two flips cancel inside lookup_volume).  K20 takes the grid as it is.
"""
import torch

from .. import lib as L
from .. import ops

_PADDING = {"zeros": 0, "border": 1, 0: 0, 1: 1, False: 0, True: 1}      # the name (forward), its index (cuda_gridsample.py:32,83), the bool the extension takes


def _padding_index(padding_mode):
    if padding_mode not in _PADDING:
        raise RuntimeError(f"padding_mode {padding_mode!r}: 'zeros' or 'border' (cuda_gridsample.py:8,13)")
    return _PADDING[padding_mode]


def _check(input, grid, dim):
    if not (input.is_cuda and grid.is_cuda):
        raise RuntimeError("gens_amd.compat.cuda_gridsample: device tensors only (no CPU path)")
    if input.dim() != dim + 2 or grid.dim() != dim + 2 or grid.shape[-1] != dim or input.shape[0] != grid.shape[0]:
        raise RuntimeError(f"expected input (N,C,{'D,' if dim == 3 else ''}H,W) and grid (N,{'Do,' if dim == 3 else ''}Ho,Wo,{dim}) with equal N")
    if input.dtype != torch.float32 or grid.dtype != torch.float32:
        raise RuntimeError("float32 tensors only (the only type the reference exercises)")


def _fast_3d(input, grid, padding, align_corners):
    """The hot path's call: K2 serves it."""
    return input.dim() == 5 and input.shape[0] == 1 and input.shape[1] % 4 == 0 and input.shape[1] > 0 and padding == 0 and bool(align_corners)


def _levels(t):
    """(1, C, D, H, W) -> list of (1, 4, D, H, W) channel groups (views, no copy for C = 4)."""
    return [t[:, c:c + 4] for c in range(0, t.shape[1], 4)]


# ---------------------------------------------------------------------------------------------------------------------------------
# K20: the general sampler as the reference's Function pair (forward / backward / backward of the backward)
# ---------------------------------------------------------------------------------------------------------------------------------
def _geom(input, grid):
    dim = grid.shape[-1]
    n, c = input.shape[0], input.shape[1]
    n_out = grid[0].numel() // dim if n > 0 else 0
    return dim, n, c, L.int_table(input.shape[2:]), n_out


def _k20_fwd(input, grid, padding, align_corners):
    dim, n, c, sizes, n_out = _geom(input, grid)
    inp, grd = input.detach().contiguous(), grid.detach().contiguous()
    out = torch.empty(n, c, *grid.shape[1:-1], device=input.device, dtype=torch.float32)
    L.call("gens_grid_sample_fwd", L.ptr(inp), L.ptr(grd), dim, n, c, sizes, n_out, padding, int(bool(align_corners)), L.ptr(out), L.stream())
    return out


def _k20_bwd(grad_output, input, grid, padding, align_corners, want_input=True, want_grid=True):
    dim, n, c, sizes, n_out = _geom(input, grid)
    go, inp, grd = grad_output.detach().contiguous(), input.detach().contiguous(), grid.detach().contiguous()
    g_in = torch.zeros_like(inp) if want_input else None
    g_grid = torch.empty_like(grd) if want_grid else None
    if want_input or want_grid:
        L.call("gens_grid_sample_bwd", L.ptr(go), L.ptr(inp), L.ptr(grd), dim, n, c, sizes, n_out, padding, int(bool(align_corners)), L.ptr(g_in),
               L.ptr(g_grid), L.stream())
    return g_in, g_grid


def _k20_bwd2(gg_input, gg_grid, grad_output, input, grid, padding, align_corners):
    dim, n, c, sizes, n_out = _geom(input, grid)
    go, inp, grd = grad_output.detach().contiguous(), input.detach().contiguous(), grid.detach().contiguous()
    ggi = None if gg_input is None else gg_input.detach().contiguous()
    ggg = gg_grid.detach().contiguous()
    gg_out = torch.empty_like(go)
    g_in = torch.zeros_like(inp)
    g_grid = torch.empty_like(grd)
    L.call("gens_grid_sample_bwd2", L.ptr(ggi), L.ptr(ggg), L.ptr(go), L.ptr(inp), L.ptr(grd), dim, n, c, sizes, n_out, padding,
           int(bool(align_corners)), L.ptr(gg_out), L.ptr(g_in), L.ptr(g_grid), L.stream())
    return gg_out, g_in, g_grid


class _GridSampleForward(torch.autograd.Function):
    """cuda_gridsample.py:21-43 (2-D) / :71-91 (3-D)."""

    @staticmethod
    def forward(ctx, input, grid, padding, align_corners):
        ctx.save_for_backward(input, grid)
        ctx.padding, ctx.align_corners = padding, align_corners
        return _k20_fwd(input, grid, padding, align_corners)

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        grad_input, grad_grid = _GridSampleBackward.apply(grad_output, input, grid, ctx.padding, ctx.align_corners)
        return grad_input, grad_grid, None, None


class _GridSampleBackward(torch.autograd.Function):
    """cuda_gridsample.py:45-66 (2-D) / :94-123 (3-D): forward = the aten backward, backward = grad2_2d / grad2_3d."""

    @staticmethod
    def forward(ctx, grad_output, input, grid, padding, align_corners):
        ctx.save_for_backward(grad_output, input, grid)
        ctx.padding, ctx.align_corners = padding, align_corners
        want_input, want_grid = ctx.needs_input_grad[1], ctx.needs_input_grad[2]          # the output_mask of the 1.11 API (:50, :99)
        g_in, g_grid = _k20_bwd(grad_output, input, grid, padding, align_corners, want_input, want_grid)
        # (ATen returns an undefined tensor for a masked-out output; autograd wants tensors here and never looks at these)
        return (g_in if g_in is not None else torch.zeros((), device=input.device).expand_as(input),
                g_grid if g_grid is not None else torch.zeros((), device=input.device).expand_as(grid))

    @staticmethod
    def backward(ctx, grad2_grad_input, grad2_grad_grid):
        grad_output, input, grid = ctx.saved_tensors
        if grad2_grad_grid is None:                                                      # (only the input's gradient was differentiated)
            grad2_grad_grid = torch.zeros_like(grid)
        gg_out, g_in, g_grid = _k20_bwd2(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, ctx.padding, ctx.align_corners)
        return gg_out, g_in, g_grid, None, None


# ---------------------------------------------------------------------------------------------------------------------------------
# the reference's names
# ---------------------------------------------------------------------------------------------------------------------------------
def grid_sample_2d(input, grid, padding_mode="zeros", align_corners=True):
    padding = _padding_index(padding_mode)
    _check(input, grid, 2)
    return _GridSampleForward.apply(input, grid, padding, bool(align_corners))


def grid_sample_3d(input, grid, padding_mode="zeros", align_corners=True):
    padding = _padding_index(padding_mode)
    _check(input, grid, 3)
    if _fast_3d(input, grid, padding, align_corners):
        pts = grid.reshape(-1, 3).flip(-1)                                # (x, y, z) of F.grid_sample -> (d, h, w) order of the volume axes
        out = ops.lookup_volume(pts, [lv.contiguous() for lv in _levels(input)])          # (N, C)
        return out.t().reshape(1, input.shape[1], *grid.shape[1:4])
    return _GridSampleForward.apply(input, grid, padding, bool(align_corners))


def grad2_2d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode=False, align_corners=True):
    """-> [grad_grad_output like grad_output, grad_input like input, grad_grid like grid]  (gridsample_cuda.cpp:26-40)."""
    padding = _padding_index(padding_mode)
    _check(input, grid, 2)
    return list(_k20_bwd2(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding, bool(align_corners)))


def grad2_3d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode=False, align_corners=True):
    """-> [grad_grad_output like grad_output, grad_input like input, grad_grid like grid]  (gridsample_cuda.cpp:42-56)."""
    padding = _padding_index(padding_mode)
    _check(input, grid, 3)
    if not _fast_3d(input, grid, padding, align_corners):
        return list(_k20_bwd2(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding, bool(align_corners)))
    c, n = input.shape[1], grid.numel() // 3
    f32 = torch.float32
    pts = grid.reshape(n, 3).flip(-1).contiguous().to(f32)
    go = grad_output.reshape(c, n).t().contiguous().to(f32)                          # (N, C), level-major like lookup_volume's output
    ggp = grad2_grad_grid.reshape(n, 3).flip(-1).contiguous().to(f32)
    levels = [lv.contiguous().to(f32) for lv in _levels(input)]
    vs = ops.VolumeSet([lv[0] for lv in levels], L.LAYOUT_PLANAR)
    ggv = None if grad2_grad_input is None else [lv.contiguous().to(f32)[0] for lv in _levels(grad2_grad_input)]
    g_in = [torch.zeros_like(lv[0]) for lv in levels]
    gg_out = torch.empty(n, c, device=input.device, dtype=f32)
    g_pts = torch.empty(n, 3, device=input.device, dtype=f32)
    L.call("gens_lookup_volume_bwd2", vs.table, vs.dim_table, vs.n, L.LAYOUT_PLANAR, L.ptr(pts), L.ptr(go), L.ptr(ggp), L.ptr_table(ggv), n,
           L.ptr(gg_out), L.ptr_table(g_in), L.ptr(g_pts), L.stream())
    return [gg_out.t().reshape(grad_output.shape), torch.cat(g_in, 0)[None], g_pts.flip(-1).reshape(grid.shape)]
