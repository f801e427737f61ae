"""K21 / K22: the depth-wise 2-D convolutions and the training-mode BatchNorm2d [+ ReLU] of the MnasNet trunk
(feature_network_mnasnet.py:53-103 -> torchvision's MNASNet layers).

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


# ------------------------------------------------------------------------------------------------------------------
# K22  BatchNorm2d in training mode [+ ReLU]
# ------------------------------------------------------------------------------------------------------------------
def batchnorm_supported(x, bn):
    """Training-mode nn.BatchNorm2d on a float32 device tensor with running statistics and a fixed momentum: what K22 covers."""
    return (x.is_cuda and x.dtype == _f32 and x.dim() == 4 and x.shape[0] * x.shape[2] * x.shape[3] > 1 and bn.training and bn.track_running_stats
            and bn.momentum is not None and x.shape[1] <= 65535 and x.shape[0] <= 1024)


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, relu):
        xc = _c(x.detach())
        n, c, h, w = xc.shape
        hw = h * w
        y = torch.empty_like(xc)
        mean_rstd = torch.empty(c, 2, device=x.device, dtype=_f32)
        scratch = torch.empty(L.load().gens_batchnorm2d_scratch_doubles(n, c, hw), device=x.device, dtype=torch.float64)
        wd, bd = (None if weight is None else _c(weight.detach())), (None if bias is None else _c(bias.detach()))
        L.call("gens_batchnorm2d_train_fwd", L.ptr(xc), L.ptr(wd), L.ptr(bd), n, c, hw, float(eps), float(momentum), 1 if relu else 0, L.ptr(y),
               L.ptr(mean_rstd), L.ptr(running_mean), L.ptr(running_var), L.ptr(num_batches_tracked, torch.int64), L.ptr(scratch, torch.float64),
               L.stream(), nbytes=12 * xc.numel(), label="gens_batchnorm2d")
        # the kernel moved the three buffers in place: tell autograd's version counters (anything that saved them for a backward then
        # raises instead of reading the new values silently) -- ctx.mark_dirty would force them to be returned as outputs
        for buf in (running_mean, running_var, num_batches_tracked):
            torch.autograd.graph.increment_version(buf)
        ctx.save_for_backward(xc, mean_rstd, wd, bd)
        ctx.relu = relu
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_y):
        xc, mean_rstd, wd, bd = ctx.saved_tensors
        n, c, h, w = xc.shape
        hw = h * w
        gy = _c(g_y.to(_f32))
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if not (need_x or need_w or need_b):
            return (None,) * 9
        g_x = torch.empty_like(xc) if need_x else None
        g_w = torch.empty(c, device=xc.device, dtype=_f32) if need_w else None
        g_b = torch.empty(c, device=xc.device, dtype=_f32) if need_b else None
        scratch = torch.empty(L.load().gens_batchnorm2d_scratch_doubles(n, c, hw), device=xc.device, dtype=torch.float64)
        L.call("gens_batchnorm2d_train_bwd", L.ptr(xc), L.ptr(gy), L.ptr(mean_rstd), L.ptr(wd), L.ptr(bd), n, c, hw, 1 if ctx.relu else 0, L.ptr(g_x),
               L.ptr(g_w), L.ptr(g_b), L.ptr(scratch, torch.float64), L.stream(), nbytes=20 * xc.numel(), label="gens_batchnorm2d")
        return g_x, g_w, g_b, None, None, None, None, None, None


def batchnorm2d_train(x, bn, relu):
    """nn.BatchNorm2d `bn` in training mode on x (n, c, h, w) [followed by ReLU]: batch statistics, running statistics and num_batches_tracked
    updated in place, differentiable in x, weight and bias."""
    return _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps, bool(relu))
