"""K15 / K16: the 3 x 3 x 3 convolutions and the instance norm + ReLU of the cost-volume U-Net.

Part of gens_amd.ops (see ops/__init__.py); citations are relative to /root/reference."""
from .base import *  # noqa: F401,F403

# ------------------------------------------------------------------------------------------------------------------
# K15  3 x 3 x 3 convolutions of the cost-volume U-Net (reg_network.py:7-50), forward / data gradient / weight gradient
# ------------------------------------------------------------------------------------------------------------------
def _conv_pad_last(t, block):
    n = t.shape[-1]
    m = (n + block - 1) // block * block
    return _c(t) if m == n else _c(torch.nn.functional.pad(t, (0, m - n)))


def _conv_gather(q, w_abt, bias, stride, reverse=False):
    """P = gather(Q): q (cq, sX, sY, sZ).  w_abt (A, B, 27): output channels A over input channels B -- or, with `reverse`, the
    stride-1 scatter through the same taps: output channels B over input channels A, taps reversed."""
    wt = w_abt.flip(-1).permute(0, 2, 1) if reverse else w_abt.permute(1, 2, 0)          # (in, 27, out)
    cq, cp = wt.shape[0], wt.shape[2]
    assert q.shape[0] == cq and all(d % stride == 0 for d in q.shape[1:])
    dims = [d // stride for d in q.shape[1:]]
    wt = _conv_pad_last(wt, 8 if cp > 4 else 4)
    p = torch.empty(cp, *dims, device=q.device, dtype=_f32)
    n = p[0].numel()
    L.call("gens_conv3d_gather", L.ptr(q, align=16), L.ptr(wt), L.ptr(None if bias is None else _c(bias.detach().to(_f32))), cp, cq, L.int_table(dims), stride,
           L.ptr(p), L.stream(), nbytes=4 * (q.numel() + p.numel()), flops=2 * 27 * cp * cq * n)
    return p


def _conv_scatter2(p, w_abt):
    """Q = scatter(P), stride 2: p (cp, X, Y, Z), w_abt (cp, cq, 27) -> q (cq, 2X, 2Y, 2Z)."""
    cp, cq = w_abt.shape[:2]
    assert p.shape[0] == cp
    dims = list(p.shape[1:])
    wt = _conv_pad_last(w_abt.permute(0, 2, 1), 8 if cq > 4 else 4)                      # (cp, 27, cq padded)
    q = torch.empty(cq, *[2 * d for d in dims], device=p.device, dtype=_f32)
    L.call("gens_conv3d_scatter2", L.ptr(p, align=16), L.ptr(wt), cp, cq, L.int_table(dims), L.ptr(q), L.stream(),
           nbytes=4 * (q.numel() + p.numel()), flops=2 * 27 * cp * cq * p[0].numel())
    return q


def _conv_wgrad(p, q, stride):
    """dW (cp, cq, 27) = sum over the voxels o of P of P[a][o] Q[b][s o + t - 1]."""
    cp, cq = p.shape[0], q.shape[0]
    dims = L.int_table(p.shape[1:])
    parts = L.load().gens_conv3d_wgrad_parts_strided(cp, cq, dims, stride)
    cpp, cqp = (cp + 3) // 4 * 4, (cq + 7) // 8 * 8
    ws = torch.empty(parts, cpp, cqp, 27, device=p.device, dtype=_f32)
    L.call("gens_conv3d_wgrad", L.ptr(p), L.ptr(q), cp, cq, dims, stride, L.ptr(ws), L.stream(),
           nbytes=4 * (q.numel() + p.numel()), flops=2 * 27 * cp * cq * p[0].numel())
    return ws.sum(0)[:cp, :cq]


def _plane_sums(x2):
    """Per-row sums of a (c, n) float32 tensor through K16's statistics pass (float64 accumulation, the whole chip per row)."""
    c, n = x2.shape
    part = torch.empty(c, L.load().gens_instnorm_blocks(c, n), 2, device=x2.device, dtype=torch.float64)
    L.call("gens_instnorm_stats", L.ptr(x2, align=16), c, n, L.ptr(part, torch.float64), L.stream(), nbytes=4 * c * n)
    return part[:, :, 0].sum(1).to(_f32)


class _Conv3d(torch.autograd.Function):
    """torch.nn.functional.conv3d(x, w, b, stride, padding=1) for a 3 x 3 x 3 kernel and batch 1."""

    @staticmethod
    def forward(ctx, x, w, b, stride):
        x3 = aligned16(x.detach()[0].to(_f32))
        ctx.save_for_backward(x3, w)
        ctx.stride, ctx.has_bias = stride, b is not None
        return _conv_gather(x3, w.detach().to(_f32).reshape(w.shape[0], w.shape[1], 27), b, stride)[None]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x3, w = ctx.saved_tensors
        g3 = _c(gy[0].to(_f32))
        w3 = w.detach().to(_f32).reshape(w.shape[0], w.shape[1], 27)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (_conv_gather(g3, w3, None, 1, reverse=True) if ctx.stride == 1 else _conv_scatter2(g3, w3))[None]
        if ctx.needs_input_grad[1]:
            gw = _conv_wgrad(g3, x3, ctx.stride).reshape(w.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _plane_sums(g3.reshape(g3.shape[0], -1))                # (aten::sum over a (4, 256^3) tensor takes 3.6 ms: 4 outputs, no parallelism)
        return gx, gw, gb, None


class _ConvTranspose3d(torch.autograd.Function):
    """torch.nn.functional.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1) for a 3 x 3 x 3 kernel, batch 1, no bias."""

    @staticmethod
    def forward(ctx, x, w):
        x3 = aligned16(x.detach()[0].to(_f32))
        ctx.save_for_backward(x3, w)
        return _conv_scatter2(x3, w.detach().to(_f32).reshape(w.shape[0], w.shape[1], 27))[None]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x3, w = ctx.saved_tensors
        g3 = _c(gy[0].to(_f32))
        w3 = w.detach().to(_f32).reshape(w.shape[0], w.shape[1], 27)
        gx = _conv_gather(g3, w3, None, 2)[None] if ctx.needs_input_grad[0] else None
        gw = _conv_wgrad(x3, g3, 2).reshape(w.shape) if ctx.needs_input_grad[1] else None
        return gx, gw


def conv3d(x, weight, bias=None, stride=1):
    """3 x 3 x 3 convolution, padding 1, stride 1 or 2: x (1, cin, X, Y, Z) -> (1, cout, X / s, Y / s, Z / s)."""
    assert x.dim() == 5 and x.shape[0] == 1 and tuple(weight.shape[2:]) == (3, 3, 3) and weight.shape[1] == x.shape[1] and stride in (1, 2)
    return _Conv3d.apply(x, weight, bias, stride)


def conv_transpose3d(x, weight):
    """3 x 3 x 3 transposed convolution, stride 2, padding 1, output_padding 1: x (1, cin, X, Y, Z) -> (1, cout, 2X, 2Y, 2Z)."""
    assert x.dim() == 5 and x.shape[0] == 1 and tuple(weight.shape[2:]) == (3, 3, 3) and weight.shape[0] == x.shape[1]
    return _ConvTranspose3d.apply(x, weight)


# ------------------------------------------------------------------------------------------------------------------
# K16  InstanceNorm3d (no affine) + ReLU of the U-Net blocks (reg_network.py:16-17,39-40)
# ------------------------------------------------------------------------------------------------------------------
_f64 = torch.float64


class _InstNormRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, skip=None):
        x2 = aligned16(x.detach().to(_f32)).reshape(x.shape[1], -1)
        c, n = x2.shape
        blocks = L.load().gens_instnorm_blocks(c, n)
        part = torch.empty(c, blocks, 2, device=x.device, dtype=_f64)
        L.call("gens_instnorm_stats", L.ptr(x2, align=16), c, n, L.ptr(part, _f64), L.stream(), nbytes=4 * c * n)
        mr = torch.empty(c, 2, device=x.device, dtype=_f32)                       # (mean, 1 / sqrt(biased variance + eps)), from float64 sums
        L.call("gens_instnorm_finish", L.ptr(part, _f64), c, n, float(eps), 0, L.ptr(mr), L.stream())
        y = torch.empty_like(x2)
        if skip is None:
            L.call("gens_instnorm_relu_fwd", L.ptr(x2), L.ptr(mr), c, n, L.ptr(y), L.stream(), nbytes=8 * c * n)
        else:
            assert skip.shape == x.shape
            L.call("gens_instnorm_relu_add_fwd", L.ptr(x2), L.ptr(mr), L.ptr(_c(skip.detach().to(_f32))), c, n, L.ptr(y), L.stream(), nbytes=12 * c * n)
        ctx.save_for_backward(x2, mr)
        ctx.blocks = blocks
        return y.reshape(x.shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x2, mr = ctx.saved_tensors
        c, n = x2.shape
        g2 = _c(gy.to(_f32)).reshape(c, n)
        part = torch.empty(c, ctx.blocks, 2, device=x2.device, dtype=_f64)
        L.call("gens_instnorm_relu_bwd_stats", L.ptr(x2), L.ptr(g2), L.ptr(mr), c, n, L.ptr(part, _f64), L.stream(), nbytes=8 * c * n)
        m12 = torch.empty(c, 2, device=x2.device, dtype=_f32)
        L.call("gens_instnorm_finish", L.ptr(part, _f64), c, n, 0.0, 1, L.ptr(m12), L.stream())
        gx = torch.empty_like(x2)
        L.call("gens_instnorm_relu_bwd", L.ptr(x2), L.ptr(g2), L.ptr(mr), L.ptr(m12), c, n, L.ptr(gx), L.stream(), nbytes=12 * c * n)
        return gx.reshape(gy.shape), None, (gy if ctx.needs_input_grad[2] else None)


def instnorm_relu(x, eps=1e-5, skip=None):
    """relu(instance_norm(x)) [+ skip] for x (n, c, ...): statistics per (sample, channel) plane, biased variance, no affine parameters
    (nn.InstanceNorm3d / nn.InstanceNorm2d in their default form followed by nn.ReLU).  A batch is n * c planes of one sample."""
    assert x.dim() >= 3
    if x.shape[0] != 1:
        shape = x.shape
        flat = (1, shape[0] * shape[1]) + tuple(shape[2:])
        return _InstNormRelu.apply(x.reshape(flat), float(eps), None if skip is None else skip.reshape(flat)).reshape(shape)
    return _InstNormRelu.apply(x, float(eps), skip)


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]      # private helpers travel too: the package namespace is the old module's
