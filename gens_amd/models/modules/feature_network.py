"""2-D feature pyramid network (last row): the reference's `FeatureNetwork`
(/root/reference/models/modules/feature_network_mnasnet.py:53-103) -- an MnasNet-1.0 trunk cut into five resolution stages (1/2 ... 1/32),
a transposed-convolution decoder with skip additions, and one 3x3 head per level -- with the same parameter names
(`layer{1..5}.*`, `decod_layer{1..5}.conv.weight`, `out_layer{1..5}.weight`), so a reference checkpoint loads with `strict=True`.

The reference takes the trunk from torchvision (`models.mnasnet1_0(pretrained=True).layers`, :57; torchvision==0.14.1 is a dependency that
is not under /root/reference and not installed here).  `_mnasnet_trunk` restates the published MnasNet-B1 depth-multiplier-1.0
architecture in torchvision's module layout (a flat `layers` Sequential: stem conv 3->32 /2, depthwise-separable 32->16, then six stacks of
inverted residuals (out, kernel, stride, expansion, repeats) = (24,3,2,3,3) (40,5,2,3,3) (80,5,2,6,3) (96,3,1,6,2) (192,5,2,6,4)
(320,3,1,6,1); BatchNorm momentum 1 - 0.9997); the channel counts are the ones the reference's decoder is written against
(:65-69: 320 -> 96 -> 40 -> 24 -> 16 -> 8).  Parity of the trunk against torchvision itself is UNPINNED (no torchvision in this image);
the wiring around it is pinned by tests/golden/g16_backbones.npz.  Weights: random initialisation (there is no network for the ImageNet
checkpoint); training / validation load theirs from the run's checkpoint as the reference does (runner.py:80).

Dense convolutions: MIOpen's territory -- the module exists so that `GenS(confs)` runs without the reference tree on sys.path
(gens_amd.models.gens._backbone).  The DEPTH-WISE convolutions are the exception: MIOpen has no tuned solver for them on gfx950 and runs its
naive kernels (3.9 ms of a 41 ms training step); `DepthwiseConv2d` is an nn.Conv2d (same parameter, same state-dict key) whose forward goes to
K21 (gens_depthwise_conv2d_*, csrc/k21_depthwise.hip) whenever the call is one K21 covers -- float32 on the device, k in {3, 5}, stride in
{1, 2}, padding k // 2 -- and to nn.Conv2d's own forward otherwise (CPU tensors, other dtypes: PyTorch's convolution, as before)."""
import os

import torch
import torch.nn as nn

_BN_MOMENTUM = 1 - 0.9997


class DepthwiseConv2d(nn.Conv2d):
    """nn.Conv2d(c, c, k, padding=k // 2, stride=s, groups=c, bias=False) on K21 where it applies."""

    use_k21 = os.environ.get("GENS_NO_K21") is None          # (switch for A/B measurements and for the test that compares the two paths)

    def forward(self, x):
        from ... import ops
        if self.use_k21 and ops.depthwise_supported(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups):
            return ops.depthwise_conv2d(x, self.weight, self.stride[0])
        return super().forward(x)


class BatchNorm2dReLU(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state-dict keys) that also applies the ReLU which follows it in the trunk when `relu` is set.
    In training mode on the device the pair is K22 (gens_batchnorm2d_train_*, csrc/k22_batchnorm.hip: two launches forward, two backward,
    the running statistics and the batch counter updated by the kernel); otherwise nn.BatchNorm2d's own forward, and the activation is
    left to the module that follows.

    The nn.ReLU slot of the Sequential keeps a REAL activation (`ReLUAfterNorm`): a forward of this class that applied the ReLU itself says so
    through `_relu_done`, which the activation consumes and then passes its input through.  A utility that replaces `_BatchNorm` modules
    (nn.SyncBatchNorm.convert_sync_batchnorm for multi-GPU training, batch-norm folding, quantisation preparation) installs a plain norm that
    never raises the flag -- the activation then simply runs; ReLU is idempotent, so no combination of the two can drop or double it."""

    use_k22 = os.environ.get("GENS_NO_K22") is None

    def __init__(self, num_features, relu=False, **kw):
        super().__init__(num_features, **kw)
        self.fused_relu = relu
        self._relu_done = False

    def forward(self, x):
        from ... import ops
        if self.use_k22 and ops.batchnorm_supported(x, self):
            self._relu_done = bool(self.fused_relu)
            return ops.batchnorm2d_train(x, self, self.fused_relu)
        self._relu_done = False
        return super().forward(x)


class ReLUAfterNorm(nn.ReLU):
    """The trunk's nn.ReLU behind a BatchNorm2dReLU: skips itself exactly when that norm's forward has just applied the activation."""

    def __init__(self, norm):
        super().__init__(inplace=True)                       # (torchvision's trunk: nn.ReLU(inplace=True))
        object.__setattr__(self, "_norm", norm)             # not a sub-module: the norm stays registered once, under its own key

    def forward(self, x):
        norm = self._norm
        if getattr(norm, "_relu_done", False):
            norm._relu_done = False
            return x
        return super().forward(x)


def _bn_relu(c):
    """BatchNorm2d + ReLU as two Sequential slots (torchvision's layout: `... .1` the norm, `... .2` the activation)."""
    norm = BatchNorm2dReLU(c, relu=True, momentum=_BN_MOMENTUM)
    return norm, ReLUAfterNorm(norm)


class _InvertedResidual(nn.Module):
    """1x1 expand -> k x k depthwise (stride) -> 1x1 project, BatchNorm after each, ReLU after the first two; identity skip when the
    shape is kept.  `layers` is torchvision's attribute name (state-dict keys `....layers.{0,1,3,4,6,7}.*`)."""

    def __init__(self, cin, cout, kernel, stride, expansion):
        super().__init__()
        mid = cin * expansion
        self.apply_residual = cin == cout and stride == 1
        self.layers = nn.Sequential(
            nn.Conv2d(cin, mid, 1, bias=False), *_bn_relu(mid),
            DepthwiseConv2d(mid, mid, kernel, padding=kernel // 2, stride=stride, groups=mid, bias=False), *_bn_relu(mid),
            nn.Conv2d(mid, cout, 1, bias=False), BatchNorm2dReLU(cout, momentum=_BN_MOMENTUM))

    def forward(self, x):
        return self.layers(x) + x if self.apply_residual else self.layers(x)


def _stack(cin, cout, kernel, stride, expansion, repeats):
    return nn.Sequential(_InvertedResidual(cin, cout, kernel, stride, expansion),
                         *[_InvertedResidual(cout, cout, kernel, 1, expansion) for _ in range(repeats - 1)])


def _mnasnet_trunk():
    """The first 14 children of torchvision's `MNASNet(alpha=1.0).layers` (the reference uses [0:14], :59-63)."""
    layers = [
        nn.Conv2d(3, 32, 3, padding=1, stride=2, bias=False), *_bn_relu(32),
        DepthwiseConv2d(32, 32, 3, padding=1, stride=1, groups=32, bias=False), *_bn_relu(32),
        nn.Conv2d(32, 16, 1, padding=0, stride=1, bias=False), BatchNorm2dReLU(16, momentum=_BN_MOMENTUM),
        _stack(16, 24, 3, 2, 3, 3), _stack(24, 40, 5, 2, 3, 3), _stack(40, 80, 5, 2, 6, 3),
        _stack(80, 96, 3, 1, 6, 2), _stack(96, 192, 5, 2, 6, 4), _stack(192, 320, 3, 1, 6, 1)]
    for m in layers:
        for c in m.modules():
            if isinstance(c, nn.Conv2d):
                nn.init.kaiming_normal_(c.weight, mode="fan_out", nonlinearity="relu")
    return layers


class _Deconv2d(nn.Module):
    """ConvTranspose2d (no bias, x2) -> InstanceNorm2d (no affine) -> ReLU (:29-50)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False)
        self.bn = nn.InstanceNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)

    use_k16 = os.environ.get("GENS_NO_K16_2D") is None

    def forward(self, x, skip=None):
        """relu(instance_norm(deconv(x))) [+ skip]: on the device the norm, the ReLU and the decoder's skip addition are K16 (gens_instnorm_*, the
        kernels of the 3-D U-Net's blocks: a batch of n * c planes), two streaming passes each way instead of aten's batch-norm route + ReLU + add."""
        y = self.conv(x)
        if self.use_k16 and y.is_cuda and y.dtype == torch.float32 and not self.bn.affine and not self.bn.track_running_stats:
            from ... import ops
            return ops.instnorm_relu(y, self.bn.eps, skip)
        y = self.relu(self.bn(y))
        return y if skip is None else y + skip


class FeatureNetwork(nn.Module):
    trunk_factory = staticmethod(_mnasnet_trunk)

    def __init__(self, confs):
        super().__init__()
        d_out = confs.get_list("d_out")
        t = self.trunk_factory()
        self.layer1 = nn.Sequential(*t[0:8])           # 1/2,  16 ch
        self.layer2 = nn.Sequential(*t[8:9])           # 1/4,  24
        self.layer3 = nn.Sequential(*t[9:10])          # 1/8,  40
        self.layer4 = nn.Sequential(*t[10:12])         # 1/16, 96
        self.layer5 = nn.Sequential(*t[12:14])         # 1/32, 320
        self.decod_layer5 = _Deconv2d(320, 96)
        self.decod_layer4 = _Deconv2d(96, 40)
        self.decod_layer3 = _Deconv2d(40, 24)
        self.decod_layer2 = _Deconv2d(24, 16)
        self.decod_layer1 = _Deconv2d(16, 8)
        self.out_layer5 = nn.Conv2d(96, d_out[4], 3, 1, 1, bias=False)
        self.out_layer4 = nn.Conv2d(40, d_out[3], 3, 1, 1, bias=False)
        self.out_layer3 = nn.Conv2d(24, d_out[2], 3, 1, 1, bias=False)
        self.out_layer2 = nn.Conv2d(16, d_out[1], 3, 1, 1, bias=False)
        self.out_layer1 = nn.Conv2d(8, d_out[0], 3, 1, 1, bias=False)

    def forward(self, x):
        """x (nv, 3, h, w), h and w multiples of 32 -> five maps (nv, d_out[i], h >> i, w >> i), finest first."""
        e1 = self.layer1(x)
        e2 = self.layer2(e1)
        e3 = self.layer3(e2)
        e4 = self.layer4(e3)
        e5 = self.layer5(e4)
        d5 = self.decod_layer5(e5, e4)
        d4 = self.decod_layer4(d5, e3)
        d3 = self.decod_layer3(d4, e2)
        d2 = self.decod_layer2(d3, e1)
        d1 = self.decod_layer1(d2)
        return [self.out_layer1(d1), self.out_layer2(d2), self.out_layer3(d3), self.out_layer4(d4), self.out_layer5(d5)]
