"""Mirror of the reference's models/modules/sdf_network.py (SDFNetwork :27-154).

The MLP itself stays PyTorch (dense GEMMs -> rocBLAS/hipBLASLt); what changes is its
volume conditioning, which goes through the fused multi-level HIP look-up with first and second derivatives
(gens_amd.ops.lookup_volume).  Parameter names (lin{l}.weight_g / weight_v / bias) match the reference so public
checkpoints load.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .embedder import get_embedder
from .linear import Linear
from .projector import lookup_volume


def _geometric_init(lin, layer, n_lin, in_dim, out_dim, d_in_embedded, feat_ch, multires, skip_in, bias, inside_outside):
    """Sphere initialisation of IDR/NeuS with the conditioning channels zeroed (sdf_network.py:63-88)."""
    with torch.no_grad():
        if layer == n_lin - 1:
            sign = -1.0 if inside_outside else 1.0
            nn.init.normal_(lin.weight, mean=sign * np.sqrt(np.pi) / np.sqrt(in_dim), std=0.0001)
            nn.init.constant_(lin.bias, -sign * bias)
            lin.weight[:, -feat_ch:] = 0.0
            lin.bias[-feat_ch:] = 0.0
            return
        nn.init.constant_(lin.bias, 0.0)
        std = np.sqrt(2) / np.sqrt(out_dim)
        if multires > 0 and layer == 0:
            nn.init.constant_(lin.weight[:, 3:], 0.0)
            nn.init.normal_(lin.weight[:, :3], 0.0, std)
        elif multires > 0 and layer in skip_in:
            nn.init.normal_(lin.weight, 0.0, std)
            lin.weight[:, -(d_in_embedded - 3 + feat_ch):] = 0.0
        else:
            nn.init.normal_(lin.weight, 0.0, std)
            lin.weight[:, -feat_ch:] = 0.0


class SDFNetwork(nn.Module):
    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=0, bias=0.5, scale=1, geometric_init=True,
                 weight_norm=True, inside_outside=False, feat_channels=32, feat_multires=2):
        super().__init__()
        self.init_feat_channels = feat_channels
        self.embed_fn_fine = None
        if multires > 0:
            self.embed_fn_fine, d_in = get_embedder(multires, input_dims=d_in)
        self.embed_fn_feat = None
        if feat_multires > 0:
            self.embed_fn_feat, feat_channels = get_embedder(feat_multires, input_dims=feat_channels)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        widths = [d_in] + [d_hidden + feat_channels] * n_layers + [d_out]
        self.num_layers = len(widths)
        n_lin = self.num_layers - 1
        for l in range(n_lin):
            out_dim = widths[l + 1]
            if l + 1 in self.skip_in:
                out_dim -= widths[0]            # room for the re-injected positional encoding
            if l < n_lin - 1:
                out_dim -= feat_channels        # room for the volume features concatenated before every hidden layer
            lin = Linear(widths[l], out_dim)
            if geometric_init:
                _geometric_init(lin, l, n_lin, widths[l], out_dim, widths[0], feat_channels, multires, self.skip_in, bias, inside_outside)
            if weight_norm:
                lin = nn.utils.weight_norm(lin)
            setattr(self, f"lin{l}", lin)
        self.activation = nn.Softplus(beta=100)

    def _hidden(self, inputs, volumes):
        """Input of the last layer: (N, d_hidden + feat_channels)."""
        feats = lookup_volume(inputs.clone(), volumes)
        if self.embed_fn_feat is not None:
            feats = self.embed_fn_feat(feats)
        pe = inputs * self.scale
        if self.embed_fn_fine is not None:
            pe = self.embed_fn_fine(pe)
        x = pe
        n_lin = self.num_layers - 1
        for l in range(n_lin - 1):
            if l in self.skip_in:
                x = torch.cat([x, pe], -1) / math.sqrt(2)
            if l > 0:
                x = torch.cat([x, feats], -1)
            x = self.activation(getattr(self, f"lin{l}")(x))
        l = n_lin - 1
        if l in self.skip_in:
            x = torch.cat([x, pe], -1) / math.sqrt(2)
        if l > 0:
            x = torch.cat([x, feats], -1)
        return x

    def forward(self, inputs, volumes):
        """inputs (N,3) world points, volumes: list of (1,4,X,Y,Z) / packed VolumeSet -> (N, d_out)."""
        x = getattr(self, f"lin{self.num_layers - 2}")(self._hidden(inputs, volumes))
        return torch.cat([x[:, :1] / self.scale, x[:, 1:]], -1)

    def sdf(self, x, volumes):
        """forward(x)[:, :1] (sdf_network.py:125-126) without the 128 feature columns nobody reads: only row 0 of the last layer is
        evaluated (its weight-normed row is the same g v / |v| the full layer forms), and no slice / cat / slice of an (N, 129) tensor
        enters the autograd graph the training step differentiates three times."""
        lin = getattr(self, f"lin{self.num_layers - 2}")
        h = self._hidden(x, volumes)
        if hasattr(lin, "weight_g"):
            w = torch._weight_norm(lin.weight_v[:1], lin.weight_g[:1], 0)
        else:
            w = lin.weight[:1]
        return torch.nn.functional.linear(h, w, lin.bias[:1]) / self.scale

    def sdf_hidden_appearance(self, x, volumes):
        return self.forward(x, volumes)

    def effective_weights(self):
        """-> ([W_0 .. W_6], [b_0 .. b_6]): the matrices the layers multiply by (weight norm applied: g v / |v| per row, the same
        `torch._weight_norm` the hook of nn.utils.weight_norm calls), with autograd history back to weight_g / weight_v / bias."""
        ws, bs = [], []
        for l in range(self.num_layers - 1):
            lin = getattr(self, f"lin{l}")
            ws.append(torch._weight_norm(lin.weight_v, lin.weight_g, 0) if hasattr(lin, "weight_g") else lin.weight)
            bs.append(lin.bias)
        return ws, bs

    def train_step(self, volumes, packed, tv_masks=None):
        """The fused training-mode evaluator (ops.SdfTrainStep: value, gradient, `smooth` and their backward in four launches) for this
        step's weights, or None when the kernels do not cover the architecture / pyramid (then the PyTorch layers run on K2 / K2'')."""
        from ... import ops
        if not (isinstance(packed, ops.VolumeSet) and ops.SdfTrainStep.supported(self, packed.n)):
            return None
        lins = [getattr(self, f"lin{l}") for l in range(self.num_layers - 1)]
        if all(hasattr(lin, "weight_g") for lin in lins) and len(lins) == 7:
            # the raw weight-normed parameters go in: the norm is one launch inside the pack, its backward rides on the gradient launch
            raw = ([lin.weight_v for lin in lins], [lin.weight_g for lin in lins], [lin.bias for lin in lins])
            return ops.SdfTrainStep(None, None, volumes, packed, raw=raw, tv_masks=tv_masks)
        ws, bs = self.effective_weights()
        return ops.SdfTrainStep(ws, bs, volumes, packed, tv_masks=tv_masks)

    @torch.enable_grad()
    def sdf_gradient_smooth(self, x, volumes):
        """-> (sdf (N,1), d sdf/dx (N,3), d(sum_k d sdf/dx_k)/dx (N,3)) from ONE forward pass.  The reference evaluates the network twice
        on the same points (implicit_surface.py:179 and, inside `gradient`, :188); the second pass reproduces the first bit for bit, so
        sharing it changes no value and removes a seventh of the training step's launches."""
        x.requires_grad_(True)
        y = self.sdf(x, volumes)
        gradients = torch.autograd.grad(y, x, torch.ones_like(y, requires_grad=False), create_graph=True, retain_graph=True, only_inputs=True)[0]
        smooth = torch.autograd.grad(gradients, x, torch.ones_like(gradients), create_graph=True, retain_graph=True, only_inputs=True)[0]
        return y, gradients, smooth

    @torch.enable_grad()
    def gradient(self, x, volumes, second_order=True):
        """-> (d sdf/dx, d(sum_k d sdf/dx_k)/dx), both (N,3), built with create_graph like sdf_network.py:131-154.

        second_order=False (results that are used detached, or inference) skips the double backward and returns (gradient, None).
        """
        if not second_order:
            x.requires_grad_(True)
            y = self.sdf(x, volumes)
            return torch.autograd.grad(y, x, torch.ones_like(y, requires_grad=False), create_graph=False, retain_graph=False)[0], None
        _, gradients, smooth = self.sdf_gradient_smooth(x, volumes)
        return gradients, smooth
