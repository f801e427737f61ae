"""nn.Conv3d / nn.ConvTranspose3d of the cost-volume U-Net with the 3 x 3 x 3 kernels of libgens_hip.so (K15) behind them: same
parameters and state-dict names as the torch modules they derive from (reg_network.py:15,38 of the reference).  Device float32 tensors
of batch 1 go to K15 (forward, data gradient, weight gradient); anything else -- the CPU golden tests of the module wiring -- is torch's
own convolution."""
import torch.nn as nn

from ... import ops


def _hip(x):
    return x.is_cuda and x.dtype == ops._f32 and x.dim() == 5 and x.shape[0] == 1


class Conv3d(nn.Conv3d):
    def forward(self, x):
        s = self.stride[0]
        if _hip(x) and self.kernel_size == (3, 3, 3) and self.padding == (1, 1, 1) and self.stride in ((1, 1, 1), (2, 2, 2)) \
                and self.dilation == (1, 1, 1) and self.groups == 1 and all(d % s == 0 for d in x.shape[2:]):
            return ops.conv3d(x, self.weight, self.bias, s)
        return super().forward(x)


class ConvTranspose3d(nn.ConvTranspose3d):
    def forward(self, x, output_size=None):
        if output_size is None and _hip(x) and self.kernel_size == (3, 3, 3) and self.padding == (1, 1, 1) and self.stride == (2, 2, 2) and self.bias is None \
                and self.output_padding == (1, 1, 1) and self.dilation == (1, 1, 1) and self.groups == 1:
            return ops.conv_transpose3d(x, self.weight)
        return super().forward(x, output_size)
