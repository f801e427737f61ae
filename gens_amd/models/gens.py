"""Drop-in for the reference's models/gens.py: same class name, constructor, methods, state-dict names and output
keys (gens.py:12-157), with the volume build and the renderer running on libgens_hip.so.

The 2-D feature CNN (MnasNet) and the 3-D regularisation U-Net are outside the accelerated path (SURVEY.md
section 2 rows 4h, 4i: dense convolutions, MIOpen through PyTorch).  In order of preference: classes
registered with `register_backbones`; the reference tree this package is dropped into
(`models.modules.feature_network_mnasnet.FeatureNetwork`, `models.modules.reg_network.RegNetwork`); this package's own
restatements with the same parameter names (`modules/feature_network.py`, `modules/reg_network.py`), so that `GenS(confs)`
also runs stand-alone.
"""
import importlib

import torch
import torch.nn as nn

from .modules.implicit_surface import ImplicitSurface
from .modules.volume import Volume

_BACKBONES = {}


def register_backbones(feature_network_cls=None, reg_network_cls=None):
    """Tell GenS which classes implement `FeatureNetwork(confs)` / `RegNetwork(confs)`."""
    if feature_network_cls is not None:
        _BACKBONES["feature"] = feature_network_cls
    if reg_network_cls is not None:
        _BACKBONES["reg"] = reg_network_cls


def _backbone(kind):
    if kind in _BACKBONES:
        return _BACKBONES[kind]
    module, name = {"feature": ("models.modules.feature_network_mnasnet", "FeatureNetwork"),
                    "reg": ("models.modules.reg_network", "RegNetwork")}[kind]
    try:
        return getattr(importlib.import_module(module), name)
    except ImportError:                    # stand-alone: this package's own restatement (same parameter names, MIOpen convolutions)
        from .modules import feature_network, reg_network
        return {"feature": feature_network.FeatureNetwork, "reg": reg_network.RegNetwork}[kind]


class GenS(nn.Module):
    def __init__(self, confs):
        super().__init__()
        self.has_vol = confs.get_bool("has_vol", default=False)
        if not self.has_vol:
            self.feature_network = _backbone("feature")(confs["feature_network"])
            self.volume = Volume(confs["volume"])
            self.reg_network = _backbone("reg")(confs["reg_network"])
            self.match_feature_network = _backbone("feature")(confs["feature_network"])
            for p in self.match_feature_network.parameters():
                p.requires_grad = False
        else:
            self.volumes = nn.ParameterList([])
            self.mask_volmes = nn.ParameterList([])       # [sic] -- the reference's spelling is part of the checkpoint format
            self.features = nn.ParameterList([])
        self.implicit_surface = ImplicitSurface(confs["implicit_surface"])

    # -- optimiser / checkpoint plumbing (gens.py:32-61) --------------------------------------------------------
    def get_optim_params(self, lr_confs):
        groups = [{"params": list(self.implicit_surface.parameters()), "lr": lr_confs["mlp_lr"]}]
        if not self.has_vol:
            groups.append({"params": list(self.feature_network.parameters()) + list(self.reg_network.parameters()), "lr": lr_confs["feat_lr"]})
        else:
            for vol, lr in zip(self.volumes, lr_confs["vol_lr"]):
                groups.append({"params": vol, "lr": lr})
        return groups

    def load_params_vol(self, path, device):
        model = torch.load(path, weights_only=False)["model"]   # ParameterLists, as saved by runner.py:332-341
        self.volumes = model["volumes"].to(device)
        self.mask_volmes = model["mask_volmes"].to(device)
        self.features = model["features"].to(device)
        self.implicit_surface.load_state_dict(model["implicit_surface"])
        self.has_vol = True

    def get_params_vol(self):
        return {"volumes": self.volumes, "mask_volmes": self.mask_volmes, "features": self.features,
                "implicit_surface": self.implicit_surface.state_dict()}

    def init_volumes(self, ipts):
        """Per-scene fine-tuning: freeze the CNN outputs into parameters (gens.py:63-85)."""
        with torch.no_grad():
            features = self.feature_network(ipts["imgs"])
            volumes, mask_volmes = self.volume.agg_mean_var(features, ipts["intrs"], ipts["c2ws"], min_vis_view=1)
            volumes = self.reg_network(volumes)
        self.volumes = nn.ParameterList([nn.Parameter(v.detach(), requires_grad=True) for v in volumes])
        self.mask_volmes = nn.ParameterList([nn.Parameter(v.detach(), requires_grad=False) for v in mask_volmes])
        self.features = nn.ParameterList([nn.Parameter(f.detach(), requires_grad=False) for f in features])
        self.has_vol = True

    # -- forward (gens.py:124-157) ------------------------------------------------------------------------------
    def forward(self, mode, ipts, cos_anneal_ratio=1.0, step=None):
        if not self.has_vol:
            imgs, intrs, c2ws = ipts["imgs"], ipts["intrs"], ipts["c2ws"]
            features = self.feature_network(imgs)
            if step is not None and step % 5 == 0:
                print("load image feature ckpt")
                self.match_feature_network.load_state_dict(self.feature_network.state_dict(), strict=True)
                for p in self.match_feature_network.parameters():
                    p.requires_grad = False
            with torch.no_grad():
                match_features = self.match_feature_network(imgs)
            volumes, mask_volmes = self.volume.agg_mean_var(features, intrs, c2ws)
            volumes = self.reg_network(volumes)
        else:
            view_ids = ipts["view_ids"] if mode != "val" else list(range(ipts["imgs"].shape[0]))
            volumes, mask_volmes = list(self.volumes), list(self.mask_volmes)
            features = [f[view_ids] for f in self.features]
            match_features = [f[view_ids] for f in self.features]
        return self.implicit_surface(mode, ipts, volumes, mask_volmes, features, match_features, cos_anneal_ratio, step)
