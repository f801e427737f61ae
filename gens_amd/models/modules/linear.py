"""nn.Linear whose parameter gradients go through the tall-operand product of libgens_hip.so (K14, gens_gemm_tn).

In a training step every layer is applied to N = 60 000 .. 250 000 rows, so `F.linear`'s backward forms dW = dY^T X and
db = dY^T 1 with a reduction length of N and a tiny result.  The BLAS library does not split that reduction over the chip: 160 us
for 128 x 61 835 x 188 (18 TFLOP/s), 0.5 ms for 32 x 247 340 x 32 (a 64 MB read), and PyTorch's column sum for the bias takes
1.2 ms on a (247 344, 33) gradient -- 8 ms of a 44 ms step in all.  Here the layer is an autograd Function whose backward calls
`ops.matmul_tn` for both; the backward is written with differentiable operations, so the second and third derivatives the SDF network
takes through its layers keep working.  Forward values are `F.linear`'s; parameter names and shapes are nn.Linear's, so checkpoints
and `weight_norm` are unaffected.  Small batches (and CPU tensors) use `F.linear` as it is."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops


class _LinearFn(torch.autograd.Function):
    """y = x w^T + b on 2-D x (the caller reshapes: a custom Function must not hand out views that are modified in place later)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = ops.matmul_nn(g, weight) if ctx.needs_input_grad[0] else None
        gw = ops.matmul_tn(g, x) if ctx.needs_input_grad[1] else None
        gb = ops.matmul_tn(g, g.new_ones(g.shape[0], 1))[:, 0] if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


class Linear(nn.Linear):
    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and x.numel() // x.shape[-1] >= ops.MATMUL_TN_MIN_ROWS and torch.is_grad_enabled():
            y = _LinearFn.apply(x.reshape(-1, x.shape[-1]), self.weight, self.bias)
            return y.reshape(*x.shape[:-1], self.out_features)
        return F.linear(x, self.weight, self.bias)
