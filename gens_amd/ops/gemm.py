"""K14: a^T b for tall operands, the weight-gradient product of the training step.

Part of gens_amd.ops (see ops/__init__.py); citations are relative to /root/reference."""
from .base import *  # noqa: F401,F403

# ------------------------------------------------------------------------------------------------------------------
# K14  a^T b for tall operands: the weight-gradient product of the training step
# ------------------------------------------------------------------------------------------------------------------
MATMUL_TN_MIN_ROWS = 8192      # below this the library GEMM is as good


def _gemm_tn(a, b):
    a, b = _c(a.detach().to(_f32)), _c(b.detach().to(_f32))
    k, m = a.shape
    n = b.shape[1]
    slabs = L.load().gens_gemm_tn_slabs(k, m, n)
    ws = torch.empty(slabs * m * n, device=a.device, dtype=_f32)
    c = torch.empty(m, n, device=a.device, dtype=_f32)
    L.call("gens_gemm_tn", L.ptr(a), L.ptr(b), k, m, n, L.ptr(ws), L.ptr(c), L.stream(), nbytes=4 * (k * (m + n) + m * n), flops=2 * k * m * n)
    return c


class _MatmulTN(torch.autograd.Function):
    """c = a^T b.  Its derivatives are tall-times-small products (_MatmulNN), whose derivatives are again a^T b products: every
    reduction over the rows stays on K14 through the second and third derivatives the SDF network takes through its layers."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _gemm_tn(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return matmul_nn(b, g.t()) if ctx.needs_input_grad[0] else None, matmul_nn(a, g) if ctx.needs_input_grad[1] else None


class _MatmulNN(torch.autograd.Function):
    """y = x w for a tall x (K, M) and a small w (M, N): the library product, with d/dw = x^T g routed to K14."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return matmul_nn(g, w.t()) if ctx.needs_input_grad[0] else None, matmul_tn(x, g) if ctx.needs_input_grad[1] else None


def _tall(a, *others):
    return a.is_cuda and a.dim() == 2 and a.shape[0] >= MATMUL_TN_MIN_ROWS and all(t.dtype == _f32 for t in (a, *others))


def matmul_tn(a, b):
    """a (K, M), b (K, N) -> a^T b (M, N).  Tall float32 device operands go to gens_gemm_tn (K split over the chip, fp32 MFMA);
    anything else to torch."""
    if _tall(a, b) and a.shape[1] <= 1024 and b.shape[1] <= 1024:
        return _MatmulTN.apply(a, b)
    return a.t() @ b


def matmul_nn(x, w):
    """x (K, M) @ w (M, N): torch's product; for tall x the gradient w.r.t. w is a K14 product."""
    if _tall(x, w) and x.shape[1] <= 1024 and w.shape[1] <= 1024 and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        return _MatmulNN.apply(x, w)
    return x @ w


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]      # private helpers travel too: the package namespace is the old module's
