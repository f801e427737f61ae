"""K21: depth-wise 2-D convolutions of the MnasNet trunk (feature_network_mnasnet.py:53-103 -> torchvision's MNASNet layers).

Part of gens_amd.ops (see ops/__init__.py); citations are relative to /root/reference."""
from .base import *  # noqa: F401,F403


def depthwise_supported(x, weight, bias, stride, padding, dilation, groups):
    """nn.Conv2d(c, c, k, padding=k//2, stride=s, groups=c, bias=False), k in {3, 5}, s in {1, 2}, float32 on the device: what K21 covers."""
    if not (x.is_cuda and x.dtype == _f32 and weight.dtype == _f32 and bias is None and x.dim() == 4 and weight.dim() == 4):
        return False
    c, k = x.shape[1], weight.shape[-1]
    return (groups == c and tuple(weight.shape) == (c, 1, k, k) and k in (3, 5) and tuple(stride) in ((1, 1), (2, 2))
            and tuple(padding) == (k // 2, k // 2) and tuple(dilation) == (1, 1) and c <= 65535)


class _DepthwiseConv2d(torch.autograd.Function):
    """forward / data gradient / weight gradient on gens_depthwise_conv2d_{fwd,dgrad,wgrad}; first order (a CNN is not differentiated twice)."""

    @staticmethod
    def forward(ctx, x, weight, stride):
        xc, wc = _c(x.detach()), _c(weight.detach())
        n, c, h, w = xc.shape
        k = wc.shape[-1]
        oh, ow = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
        out = torch.empty(n, c, oh, ow, device=x.device, dtype=_f32)
        L.call("gens_depthwise_conv2d_fwd", L.ptr(xc), L.ptr(wc), n, c, h, w, k, stride, L.ptr(out), L.stream(),
               nbytes=4 * (xc.numel() + out.numel()), label="gens_depthwise_conv2d")
        ctx.save_for_backward(xc, wc)
        ctx.stride = stride
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        xc, wc = ctx.saved_tensors
        n, c, h, w = xc.shape
        k, stride = wc.shape[-1], ctx.stride
        go = _c(g_out.to(_f32))
        g_x = g_w = None
        if ctx.needs_input_grad[0]:
            g_x = torch.empty_like(xc)
            L.call("gens_depthwise_conv2d_dgrad", L.ptr(go), L.ptr(wc), n, c, h, w, k, stride, L.ptr(g_x), L.stream(),
                   nbytes=4 * (go.numel() + g_x.numel()), label="gens_depthwise_conv2d")
        if ctx.needs_input_grad[1]:
            parts = L.load().gens_depthwise_conv2d_wgrad_parts(n, c, h, w, k, stride)
            partial = torch.empty(parts, c, k, k, device=xc.device, dtype=_f32)
            L.call("gens_depthwise_conv2d_wgrad", L.ptr(xc), L.ptr(go), n, c, h, w, k, stride, L.ptr(partial), L.stream(),
                   nbytes=4 * (go.numel() + xc.numel()), label="gens_depthwise_conv2d")
            g_w = (partial[0] if parts == 1 else partial.sum(0)).reshape(wc.shape)
        return g_x, g_w, None


def depthwise_conv2d(x, weight, stride):
    """x (n, c, h, w), weight (c, 1, k, k), padding k // 2 -> (n, c, oh, ow); differentiable in x and weight."""
    return _DepthwiseConv2d.apply(x, weight, int(stride))
