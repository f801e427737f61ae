"""3-D regularisation U-Net over the cost-volume pyramid (last row): the reference's `RegNetwork`
(/root/reference/models/modules/reg_network.py:105-169) with the same parameter names (`conv0.conv.weight`,
`encoder_layers.{i}.{0,1}.conv.weight`, `decoder_layers.{i}.conv.weight`, `out_layers.{i}.{weight,bias}`), so a reference checkpoint
loads with `strict=True`.  The module exists so that `GenS(confs)` runs without the reference tree on sys.path (gens_amd.models.gens._backbone); its 3x3x3
convolutions run on K15 (gens_amd/csrc/k15_conv3d.hip): MIOpen's backward for these few-channel 256^3 layers takes seconds.

Stage i of the encoder halves the resolution (D / 2^(i+1)) and widens to d_base * 2^i channels, then takes the next level of the
cost-volume pyramid as extra input channels; the decoder walks back up with transposed convolutions and skip additions; every level gets
its own 3x3x3 output head (finest first)."""
import torch
import torch.nn as nn

from ... import ops
from .conv3d import Conv3d, ConvTranspose3d


class _Block3d(nn.Module):
    """conv (no bias) -> InstanceNorm3d (no affine) -> ReLU; `conv`, `bn`, `relu` are the reference's attribute names (:7-27, :30-50)."""

    def __init__(self, conv):
        super().__init__()
        self.conv = conv
        self.bn = nn.InstanceNorm3d(conv.out_channels)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x, skip=None):
        """relu(norm(conv(x))) [+ skip]"""
        x = self.conv(x)
        if x.is_cuda and x.dtype == torch.float32 and x.shape[0] == 1 and not self.bn.affine and not self.bn.track_running_stats:
            return ops.instnorm_relu(x, self.bn.eps, skip)                 # K16: statistics + normalise + ReLU (+ the skip addition) in two streaming passes
        x = self.relu(self.bn(x))
        return x if skip is None else x + skip


def _conv(cin, cout, stride):
    return _Block3d(Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False))


def _deconv(cin, cout):
    return _Block3d(ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False))


class RegNetwork(nn.Module):
    def __init__(self, conf):
        super().__init__()
        d_volume = conf.get_list("d_voluem")          # [sic]: the key of confs/gens.conf
        d_base = conf.get_int("d_base")
        d_out = conf.get_list("d_out")
        self.num_stage = len(d_out)
        self.encoder_layers = nn.ModuleList()         # registration order = state-dict order of the reference (:116-120)
        self.decoder_layers = nn.ModuleList()
        self.out_layers = nn.ModuleList()
        self.conv0 = _conv(d_volume[0], d_base, 1)
        cin = d_base
        for i in range(self.num_stage):
            wide, narrow = d_base << i, d_base << max(i - 1, 0)
            self.encoder_layers.append(nn.Sequential(_conv(cin, wide, 2), _conv(wide, wide, 1)))
            if i < self.num_stage - 1:
                cin = wide + d_volume[i + 1]
            self.out_layers.append(Conv3d(narrow, d_out[i], 3, 1, 1))
            self.decoder_layers.append(_deconv(wide, narrow))

    def forward(self, volumes):
        assert len(volumes) == self.num_stage
        x = self.conv0(volumes[0])
        skips = [x]
        for i, enc in enumerate(self.encoder_layers):
            x = enc(x)
            skips.append(x)
            if i < self.num_stage - 1:
                x = torch.cat([x, volumes[i + 1]], dim=1)
        ups = []
        for i in range(self.num_stage - 1, -1, -1):
            x = self.decoder_layers[i](x, skips[i])
            ups.append(x)
        ups.reverse()                                  # finest first
        return [head(u) for head, u in zip(self.out_layers, ups)]
