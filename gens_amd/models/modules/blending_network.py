"""Mirror of the reference's models/modules/blending_network.py (IBRNet-style colour blending, :22-118).
Tiny MLPs: stays PyTorch.  Same parameter names as the reference."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .linear import Linear


def _kaiming(m):
    if isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight.data)
        if m.bias is not None:
            nn.init.zeros_(m.bias.data)


def _mlp(*widths, last=None):
    layers = []
    for i in range(len(widths) - 1):
        layers.append(Linear(widths[i], widths[i + 1]))
        if i < len(widths) - 2 or last == "elu":
            layers.append(nn.ELU(inplace=True))
    if last == "sigmoid":
        layers.append(nn.Sigmoid())
    return nn.Sequential(*layers)


class BlendingNetwork(nn.Module):
    def __init__(self, d_feature=16, anti_alias_pooling=True):
        super().__init__()
        self.anti_alias_pooling = anti_alias_pooling
        if anti_alias_pooling:
            self.s = nn.Parameter(torch.tensor(0.2), requires_grad=True)
        f = d_feature + 3
        self.ray_dir_fc = _mlp(4, 16, f, last="elu")
        self.base_fc = _mlp(3 * f, 64, 32, last="elu")
        self.vis_fc = _mlp(32, 32, 33, last="elu")
        self.vis_fc2 = _mlp(32, 32, 1, last="sigmoid")
        self.rgb_fc = _mlp(32 + 1 + 4, 16, 8, 1)
        for m in (self.base_fc, self.vis_fc2, self.vis_fc, self.rgb_fc):
            m.apply(_kaiming)

    def forward(self, rgb_feat, ray_diff, mask):
        """rgb_feat (N,S,3+C), ray_diff (N,S,4), mask (N,S) -> blended colour (N,3)."""
        mask = mask[:, :, None].to(rgb_feat.dtype)
        n_src = rgb_feat.shape[1]
        rgb_in = rgb_feat[..., :3]
        x = rgb_feat + self.ray_dir_fc(ray_diff)
        if self.anti_alias_pooling:
            e = torch.exp(torch.abs(self.s) * (ray_diff[..., 3:4] - 1))
            w = (e - e.min(dim=1, keepdim=True)[0]) * mask
        else:
            w = mask
        w = w / (w.sum(dim=1, keepdim=True) + 1e-8)
        mean = (x * w).sum(dim=1, keepdim=True)
        var = (w * (x - mean) ** 2).sum(dim=1, keepdim=True)
        h = self.base_fc(torch.cat([mean.expand(-1, n_src, -1), var.expand(-1, n_src, -1), x], -1))
        hv = self.vis_fc(h * w)
        h = h + hv[..., :-1]
        vis = torch.sigmoid(hv[..., -1:]) * mask
        vis = self.vis_fc2(h * vis) * mask
        score = self.rgb_fc(torch.cat([h, vis, ray_diff], -1)).masked_fill(mask == 0, -1e9)
        return (rgb_in * F.softmax(score, dim=1)).sum(dim=1)
