"""nn.Conv3d / nn.ConvTranspose3d of the cost-volume U-Net with the 3 x 3 x 3 kernels of libgens_hip.so (K15) behind them: same
parameters and state-dict names as the torch modules they derive from (reg_network.py:15,38 of the reference).  Device float32 tensors
of batch 1 go to K15 (forward, data gradient, weight gradient); anything else -- the CPU golden tests of the module wiring -- is torch's
own convolution."""
import warnings

import torch.nn as nn

from ... import ops

_WARNED = set()


def _hip(x):
    return x.is_cuda and x.dtype == ops._f32 and x.dim() == 5 and x.shape[0] == 1


def _fall_through(kind, module, x):
    """A DEVICE call that K15 does not cover goes to MIOpen -- whose backward for these few-channel 3-D layers takes seconds per layer
    (2.4 s against 3.4 ms) -- so say so, once per shape, instead of silently losing three orders of magnitude."""
    if x.is_cuda:
        key = (kind, tuple(x.shape), str(x.dtype), module.stride, module.padding, module.kernel_size, module.bias is not None)
        if key not in _WARNED:
            _WARNED.add(key)
            warnings.warn(f"gens_amd: {kind} input {tuple(x.shape)} {x.dtype} (stride {module.stride}, padding {module.padding}, kernel "
                          f"{module.kernel_size}, bias {module.bias is not None}) is outside what the K15 kernels cover (batch 1, float32, 3x3x3, "
                          "padding 1, stride 1 / 2 with even extents; transposed: stride 2, output_padding 1, no bias): using torch / MIOpen, "
                          "whose backward is orders of magnitude slower for these layers", RuntimeWarning, stacklevel=3)


class Conv3d(nn.Conv3d):
    def forward(self, x):
        s = self.stride[0]
        if _hip(x) and self.kernel_size == (3, 3, 3) and self.padding == (1, 1, 1) and self.stride in ((1, 1, 1), (2, 2, 2)) \
                and self.dilation == (1, 1, 1) and self.groups == 1 and all(d % s == 0 for d in x.shape[2:]):
            return ops.conv3d(x, self.weight, self.bias, s)
        _fall_through("Conv3d", self, x)
        return super().forward(x)


class ConvTranspose3d(nn.ConvTranspose3d):
    def forward(self, x, output_size=None):
        if output_size is None and _hip(x) and self.kernel_size == (3, 3, 3) and self.padding == (1, 1, 1) and self.stride == (2, 2, 2) and self.bias is None \
                and self.output_padding == (1, 1, 1) and self.dilation == (1, 1, 1) and self.groups == 1:
            return ops.conv_transpose3d(x, self.weight)
        _fall_through("ConvTranspose3d", self, x)
        return super().forward(x, output_size)
