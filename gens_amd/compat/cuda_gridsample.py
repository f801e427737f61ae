"""Drop-in for the reference's only native boundary, `models/modules/grid_sample_cuda/cuda_gridsample.py` (+ the `gridsample_grad2`
extension it JIT-builds from gridsample_cuda.cpp / gridsample_cuda.cu), in the REFERENCE'S OWN calling convention:

    grid_sample_3d(input (1,C,D,H,W), grid (1,Do,Ho,Wo,3), padding_mode='zeros', align_corners=True) -> (1,C,Do,Ho,Wo)
        twice differentiable like the reference's Function pair (cuda_gridsample.py:12-14,71-123): first backward =
        gens_lookup_volume_bwd (what aten::grid_sampler_3d_backward returns), second backward = gens_lookup_volume_bwd2 (what
        grad2_3d returns); the outputs of the second backward are constants, as in the reference.
    grad2_3d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode, align_corners) -> [ggO, gI, gG]
        the extension's entry point (gridsample_cuda.cpp:42-56) on the same kernels.

A maintainer who wants ONLY the sampler replaced (and keeps the reference's projector / networks) puts this file in place of
cuda_gridsample.py.  The grid follows F.grid_sample: last axis (x, y, z) indexes (W, H, D), i.e. input
element [d(z)][h(y)][w(x)]; the kernels read element [ix(px)][iy(py)][iz(pz)] of an (X, Y, Z) volume at point p, so a grid point is
handed over flipped (projector.py:223 flips it the other way before calling: the two flips cancel inside lookup_volume).

Restrictions (each raises, none falls back): batch 1, C a multiple of 4 (C = 4 is one kernel level; more channels are split into
4-channel levels), padding_mode 'zeros' (the only mode the reference calls, projector.py:229,238), align_corners=True, float32 device
tensors.  `grid_sample_2d` / `grad2_2d` (never called by the reference's hot path) are not provided.
"""
import torch

from .. import lib as L
from .. import ops


def _check(input, grid, padding_mode, align_corners):
    if not (input.is_cuda and grid.is_cuda):
        raise RuntimeError("gens_amd.compat.cuda_gridsample: device tensors only (no CPU path)")
    if input.dim() != 5 or grid.dim() != 5 or grid.shape[-1] != 3 or input.shape[0] != 1 or grid.shape[0] != 1:
        raise RuntimeError("expected input (1,C,D,H,W) and grid (1,Do,Ho,Wo,3)")
    if input.shape[1] % 4 != 0:
        raise RuntimeError(f"channel count {input.shape[1]} is not a multiple of 4 (the kernels read 4-channel levels)")
    if padding_mode not in ("zeros", 0, False):
        raise RuntimeError("only padding_mode='zeros' is implemented (the reference never passes 'border': projector.py:229,238)")
    if not align_corners:
        raise RuntimeError("only align_corners=True is implemented (projector.py:229,238)")


def _levels(t):
    """(1, C, D, H, W) -> list of (1, 4, D, H, W) channel groups (views, no copy for C = 4)."""
    return [t[:, c:c + 4] for c in range(0, t.shape[1], 4)]


def grid_sample_3d(input, grid, padding_mode="zeros", align_corners=True):
    _check(input, grid, padding_mode, align_corners)
    pts = grid.reshape(-1, 3).flip(-1)                                    # (x, y, z) of F.grid_sample -> (d, h, w) order of the volume axes
    out = ops.lookup_volume(pts, [lv.contiguous() for lv in _levels(input)])          # (N, C)
    return out.t().reshape(1, input.shape[1], *grid.shape[1:4])


def grad2_3d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode=False, align_corners=True):
    """-> [grad_grad_output like grad_output, grad_input like input, grad_grid like grid]  (gridsample_cuda.cpp:42-56)."""
    _check(input, grid, padding_mode, align_corners)
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
