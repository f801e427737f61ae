"""CPU oracle of the reference's sampler boundary in full generality -- TEST INFRASTRUCTURE, never the product path (gens_oracle.py's header).

What `models/modules/grid_sample_cuda/cuda_gridsample.py:7-14` exports: grid_sample_2d / grid_sample_3d with padding_mode in
{'zeros', 'border'} and either align_corners, twice differentiable (the second backward is the reference's CUDA extension,
gridsample_cuda.cu:27-210 (2-D), :212-533 (3-D), which cannot be built here).  The arithmetic itself is a third-party dependency's:
aten::grid_sampler_2d / _3d of PyTorch (pinned torch==1.13.1 in the reference's requirements.txt:17; this container's torch 2.10 keeps the
same unnormalize / clip / within-bounds / bilinear definitions, ATen/native/GridSamplerUtils.h, cuda/GridSampler.cuh).

`sample` restates it with plain differentiable torch operations (any float dtype, so float64 runs give a yardstick), which autograd then
differentiates to any order.  Pinning: value and FIRST derivatives against ATen's own F.grid_sample + autograd on the CPU for every mode
(tests/test_oracle_golden.py); second derivatives are what autograd makes of the pinned first-order function, checked against float64
central differences of ATen's backward (same test).  Corner weights are formed like ATen's ((x_se - x) * (y_se - y) ...).
"""
import itertools

import torch


def source_index(g, size, padding_mode, align_corners):
    """Normalised coordinate -> source index (GridSamplerUtils.h: grid_sampler_unnormalize + clip_coordinates for 'border')."""
    if align_corners:
        s = (g + 1) / 2 * (size - 1)
    else:
        s = ((g + 1) * size - 1) / 2
    if padding_mode == "border":
        # clip_coordinates_set_grad: outside [0, size-1] the index is pinned and its derivative with respect to g is zero
        lo, hi = torch.zeros_like(s), torch.full_like(s, float(size - 1))
        s = torch.where(s <= 0, lo, torch.where(s >= size - 1, hi, s))
    else:
        assert padding_mode == "zeros", padding_mode
    return s


def sample(input, grid, padding_mode="zeros", align_corners=True):
    """input (N, C, *spatial) with 2 or 3 spatial axes, grid (N, *out, DIM) with the last axis (x, y[, z]) -> (N, C, *out).
    Bilinear / trilinear; differentiable in input and grid to any order."""
    dim = grid.shape[-1]
    assert input.dim() == dim + 2 and grid.dim() == dim + 2 and input.shape[0] == grid.shape[0]
    n, c = input.shape[:2]
    spatial = input.shape[2:]                      # slowest first: (D,) H, W
    out_shape = grid.shape[1:-1]
    g = grid.reshape(n, -1, dim)
    flat = input.reshape(n, c, -1)
    base, frac = [], []
    for a in range(dim):                           # grid axis a indexes tensor axis (dim - 1 - a)
        size = spatial[dim - 1 - a]
        s = source_index(g[..., a], size, padding_mode, align_corners)
        f = torch.floor(s.detach())
        finite = torch.isfinite(s)
        f = torch.where(finite, f, torch.full_like(f, -10.0)).clamp(-10, size + 10)
        s = torch.where(finite, s, f)
        base.append(f)
        frac.append((s, f))
    strides = [1] * dim
    for a in range(1, dim):
        strides[a] = strides[a - 1] * spatial[dim - a]
    out = 0
    for bits in itertools.product((0, 1), repeat=dim):
        w = 1
        ok = torch.ones_like(base[0], dtype=torch.bool)
        lin = torch.zeros_like(base[0], dtype=torch.long)
        for a, bit in enumerate(bits):
            s, f = frac[a]
            size = spatial[dim - 1 - a]
            idx = f + bit
            w = w * ((s - f) if bit else ((f + 1) - s))
            ok = ok & (idx >= 0) & (idx < size)
            lin = lin + idx.clamp(0, size - 1).long() * strides[a]
        vals = torch.gather(flat, 2, lin[:, None, :].expand(n, c, -1))
        out = out + vals * (w * ok.to(input.dtype))[:, None, :]
    return out.reshape(n, c, *out_shape)
