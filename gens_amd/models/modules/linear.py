"""nn.Linear whose bias gradient is a GEMV.

`F.linear`'s backward forms grad_bias = grad_out.sum(0).  For output widths that are not a multiple of 4 (the shipped networks have
101, 33 and 23) that column reduction runs on PyTorch's scalar path: 1.2 ms for a (247 344, 33) gradient on MI355X, 2.5 ms per
training step in all, more than the layers' GEMMs.  Here the bias is added by a small autograd Function whose backward computes
ones(1, N) @ grad_out instead -- the same sums through the BLAS library.  Its backward is written with differentiable torch ops, so the
second and third derivatives the SDF network needs keep working.  Forward values are those of `F.linear` (a float32 GEMM result plus
the bias); parameter names and shapes are nn.Linear's, so checkpoints and `weight_norm` are unaffected."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _AddBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        return x + bias

    @staticmethod
    def backward(ctx, g):
        g2 = g.reshape(-1, g.shape[-1])
        return g, torch.matmul(g2.new_ones(1, g2.shape[0]), g2)[0]


class Linear(nn.Linear):
    def forward(self, x):
        if self.bias is None or self.out_features % 4 == 0 or self.out_features == 1 or not x.is_cuda:
            return F.linear(x, self.weight, self.bias)
        return _AddBias.apply(F.linear(x, self.weight), self.bias)
