"""Drop-in for the reference's models/gens.py: same class name, constructor, methods, state-dict names and output
keys (gens.py:12-157), with the volume build and the renderer running on libgens_hip.so.

The 2-D feature CNN (MnasNet) and the 3-D regularisation U-Net.  In order of preference: classes
registered with `register_backbones`; then
  * U-Net: THIS package's `modules/reg_network.RegNetwork` (same constructor, same state-dict keys -- checkpoints load with strict=True --
    on the K15 / K16 kernels: 13.8 ms forward + backward against 6.2 s through MIOpen).  The reference tree's own
    class is used only when registered explicitly: picking it silently cost three orders of magnitude in the drop-in scenario;
  * feature CNN: the reference tree this package is dropped into (`models.modules.feature_network_mnasnet.FeatureNetwork`, i.e. the real
    torchvision trunk with its pretrained weights) when importable, else this package's restatement of the same architecture.
"""
import importlib
import warnings

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
        cls = _BACKBONES[kind]
        if kind == "reg" and cls.__module__.split(".")[0] != __name__.split(".")[0]:
            warnings.warn(f"gens_amd: RegNetwork registered from {cls.__module__}: its nn.Conv3d / nn.InstanceNorm3d layers run through MIOpen "
                          "(seconds per training step at 256^3, DESIGN.md section 4c); gens_amd's own RegNetwork has the same state-dict keys",
                          RuntimeWarning, stacklevel=3)
        return cls
    from .modules import feature_network, reg_network
    if kind == "reg":
        return reg_network.RegNetwork
    try:
        return getattr(importlib.import_module("models.modules.feature_network_mnasnet"), "FeatureNetwork")
    except ImportError:                    # stand-alone: this package's own restatement (same parameter names, MIOpen convolutions)
        return feature_network.FeatureNetwork


def _check_limits(confs):
    """The kernels' hard limits, checked where the model is built (a wrong configuration used to fail deep inside a kernel call, or to drop
    silently to the unfused path)."""
    from .. import lib as L
    surf = confs["implicit_surface"]
    feat_ch = surf["sdf_network"]["feat_channels"]
    if feat_ch % 4 != 0 or not 1 <= feat_ch // 4 <= L.MAX_LEVELS:
        raise ValueError(f"implicit_surface.sdf_network.feat_channels = {feat_ch}: the look-up kernels read 4-channel volume levels, 1 to "
                         f"{L.MAX_LEVELS} of them (include/gens_hip.h: GENS_MAX_LEVELS); confs/gens.conf uses 4 channels x 5 levels")
    if not confs.get_bool("has_vol", default=False):
        d_out = confs["reg_network"]["d_out"]
        dims = confs["volume"]["volume_dims"]
        if any(c != 4 for c in d_out) or len(d_out) != len(dims) or 4 * len(dims) != feat_ch:
            raise ValueError(f"reg_network.d_out = {list(d_out)}, volume.volume_dims = {list(dims)}, feat_channels = {feat_ch}: every volume level "
                             "must have 4 channels and feat_channels must equal 4 x the number of levels (the kernels' texel layout)")
        if any(c != 4 for c in confs["feature_network"]["d_out"]) or len(confs["feature_network"]["d_out"]) > 5:
            raise ValueError("feature_network.d_out: at most 5 feature levels of 4 channels each (K1 / K4 / K7 read 4-channel texels)")
    if feat_ch // 4 > 5:
        warnings.warn(f"gens_amd: {feat_ch // 4} volume levels: the fused SDF kernels (gens_sdf_value / gens_sdf_grad / gens_sdf_train_*) are built for 1 to 5 "
                      "levels; this configuration runs the PyTorch layers on the K2 look-up kernels (correct, several times slower)", RuntimeWarning, stacklevel=3)


class GenS(nn.Module):
    def __init__(self, confs):
        super().__init__()
        _check_limits(confs)
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
        """gens.py:32-45: the parameter groups runner.py:96-97 hands to torch.optim.Adam.  On the device every group also carries `fused=True`
        (a per-group option torch's Adam honours in step()): the same update rule in ONE launch per group instead of the multi-tensor form's ~10
        passes over the parameters -- the fine-tune volumes are 307 MB, 1.3 ms per step.  GENS_FUSED_ADAM=0: the reference's groups as they are."""
        import os
        groups = [{"params": list(self.implicit_surface.parameters()), "lr": lr_confs["mlp_lr"]}]
        if not self.has_vol:
            groups.append({"params": list(self.feature_network.parameters()) + list(self.reg_network.parameters()), "lr": lr_confs["feat_lr"]})
        else:
            for vol, lr in zip(self.volumes, lr_confs["vol_lr"]):
                groups.append({"params": vol, "lr": lr})
        if os.environ.get("GENS_FUSED_ADAM", "1") not in ("0", "off", "false", "no"):
            def on_device(ps):
                ps = [ps] if torch.is_tensor(ps) else ps
                return all(p.is_cuda and p.is_floating_point() for p in ps)
            if all(on_device(g["params"]) for g in groups):
                for g in groups:
                    g["fused"] = True
        return groups

    def load_params_vol(self, path, device):
        model = torch.load(path, weights_only=False)["model"]   # ParameterLists, as saved by runner.py:332-341
        self.volumes = model["volumes"].to(device)
        self.mask_volmes = model["mask_volmes"].to(device)
        self.features = model["features"].to(device)
        self.implicit_surface.load_state_dict(model["implicit_surface"])
        self.has_vol = True
        self._drop_captured_steps()

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
        self._drop_captured_steps()

    # -- forward (gens.py:124-157) ------------------------------------------------------------------------------
    def _reload_match(self, step):
        """gens.py:131-135: every fifth epoch's first step copies the feature network into its frozen matching twin."""
        if step is not None and step % 5 == 0:
            print("load image feature ckpt")
            self.match_feature_network.load_state_dict(self.feature_network.state_dict(), strict=True)
            for p in self.match_feature_network.parameters():
                p.requires_grad = False

    def _view_index_of(self, view_ids):
        """(gens.py:151-153: `self.features[i][view_ids]` with a Python list builds an index tensor on the host and copies it over -- every step, a
        pageable host-to-device copy, and not capturable into a graph: the index tensor is kept per list of ids; one gather serves both uses,
        the reference's two copies hold the same values)"""
        if torch.is_tensor(view_ids):
            return view_ids
        ids = tuple(int(v) for v in view_ids)
        cache = getattr(self, "_view_index", None)
        if cache is None or cache[0] != ids or cache[1].device != self.features[0].device:
            cache = self._view_index = (ids, torch.tensor(ids, dtype=torch.long, device=self.features[0].device))
        return cache[1]

    def _seed_frozen_layouts(self, selected, index):
        """Fine-tuning keeps the feature pyramid of EVERY view of the scene as frozen parameters and takes this step's views out of it
        (gens.py:151-153).  The kernels read texel layouts of the maps (ops.pack_maps) and, for the patch warp, the three finest levels up-sampled
        into one map (ops.build_warp_features); both are cached on the map TENSOR -- and `features[i][view_ids]` is a new tensor every step, so every
        step re-packed and re-up-sampled what never changes (0.2 ms of a 5.5 ms step; 0.7 of 6.5 at the shipped 1152 x 1600).  Here the layouts of
        the whole frozen pyramid are built once (cached on the parameters, for their current versions) and this step's views are taken out of
        THEM: the selected maps carry ready layouts when the renderer asks."""
        from .. import ops
        from ..ops.base import kernels, pack_maps
        if not kernels.tex_cache or any(f.requires_grad for f in self.features) or not selected or selected[0].device.type != "cuda":
            return
        grad_mode = torch.is_grad_enabled()
        full = pack_maps(list(self.features))
        for f, t in zip(selected, full):
            f._gens_tex = (f._version, grad_mode and f.requires_grad, t.index_select(0, index))
        if len(selected) >= 3:
            warp_full, channels = ops.build_warp_features(list(self.features[:3]))
            selected[0]._gens_warp = (tuple((id(f), f._version) for f in selected[:3]), list(selected[1:3]), (warp_full.index_select(0, index), channels))

    def _select_frozen_views(self, index):
        """`[f[view_ids] for f in self.features]` (gens.py:151-153) together with the same selection of the frozen maps' texel / warp layouts
        (_seed_frozen_layouts) in ONE launch (gens_select_views): eleven torch.index_select launches per step otherwise, ~0.13 ms of a 5 ms step
        at 480 x 640 and six times that at the shipped 1152 x 1600.  Maps that want a gradient, or live elsewhere, take the torch operators."""
        from .. import ops
        from ..ops.base import kernels, pack_maps
        feats = list(self.features)
        if (not kernels.tex_cache or not kernels.select_views or not feats or feats[0].device.type != "cuda" or any(f.requires_grad or f.dtype != torch.float32 for f in feats)
                or index.dtype != torch.long or any((f[0].numel() & 3) or not f.is_contiguous() for f in feats)):
            features = [f.index_select(0, index) for f in feats]
            self._seed_frozen_layouts(features, index)
            return features
        from .. import lib as L
        grad_mode = torch.is_grad_enabled()
        full = pack_maps(feats)
        srcs = feats + list(full)
        warp = None
        if len(feats) >= 3:
            warp = ops.build_warp_features(feats[:3])
            srcs.append(warp[0])
        if len(srcs) > 16 or any((t[0].numel() & 3) or t.data_ptr() % 16 for t in srcs):
            features = [f.index_select(0, index) for f in feats]
            self._seed_frozen_layouts(features, index)
            return features
        n_sel = int(index.shape[0])
        dsts = [torch.empty((n_sel,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype) for t in srcs]
        L.call("gens_select_views", L.ptr_table([t.detach() for t in srcs], align=16), L.ptr_table(dsts, align=16), L.int_table([t[0].numel() for t in srcs]),
               L.int_table([t.shape[0] for t in srcs]), len(srcs), L.ptr(index, torch.long), n_sel, L.stream(),
               nbytes=8 * n_sel * sum(t[0].numel() for t in srcs))
        n = len(feats)
        features = dsts[:n]
        for f, t in zip(features, dsts[n:2 * n]):
            f._gens_tex = (f._version, grad_mode and f.requires_grad, t)
        if warp is not None:
            features[0]._gens_warp = (tuple((id(f), f._version) for f in features[:3]), list(features[1:3]), (dsts[2 * n], warp[1]))
        return features

    def _tree_modes(self):
        return tuple(m.training for m in self.modules())

    def _drop_captured_steps(self):
        """The tensors under the captured steps changed (parameters moved, volumes frozen or loaded): they can never be replayed again, and an
        entry keeps its private graph pool -- a whole step's activations, the 256^3 U-Net's included -- until it is dropped."""
        self._mode_version = getattr(self, "_mode_version", 0) + 1
        for owner in (self, getattr(self, "implicit_surface", None)):
            auto = getattr(owner, "_auto", None) if owner is not None else None
            if auto is not None:
                auto.reset()

    def train(self, mode=True):
        # A captured step belongs to one train / eval setting of the module tree: the setting is part of the step's signature.  runner.py:140 calls
        # model.train() at the start of EVERY epoch and validate() switches to eval and back around an image (runner.py:201): a call that changes
        # nothing, or that returns to a setting seen before, finds its captured step again -- no re-capture (two eager steps + a capture), no second
        # copy of the step's activations pinned in a graph pool that runner.py's empty_cache() cannot free.
        out = super().train(mode)
        self._mode_sig = hash(self._tree_modes())
        return out

    def _apply(self, fn, *args, **kwargs):
        self._drop_captured_steps()                # .to() / .cuda() / .float(): the parameters move
        return super()._apply(fn, *args, **kwargs)

    def _step_refs(self):
        """Every parameter and buffer of the model, listed once per module-tree version (walking ~500 modules costs a millisecond per call)."""
        version = (getattr(self, "_mode_version", 0), self.has_vol, len(getattr(self, "volumes", ())) if self.has_vol else 0)
        cache = getattr(self, "_refs_cache", None)
        if cache is None or cache[0] != version:
            cache = self._refs_cache = (version, list(self.parameters()) + list(self.buffers()))
        return cache[1]

    def _auto_graph_ok(self, mode, ipts, step):
        """May this call run as a captured step (gens_amd.graph.AutoGraph)?  Training / fine-tune calls on the device whose render takes the fused
        path; the one step in five epochs that refreshes the matching network stays eager (its copy sits between the two CNN passes)."""
        from .. import graph
        if mode == "val" or not getattr(self, "auto_graph", True) or not graph.auto_graph_enabled() or not torch.is_grad_enabled():
            return False
        if not self.has_vol and step is not None and step % 5 == 0:
            return False
        if self.has_vol:
            volumes, n_feat = list(self.volumes), len(self.features)
        else:
            dims = getattr(self.volume, "volume_dims", None)
            if dims is None:
                return False
            volumes, n_feat = None, len(dims)
        surf = self.implicit_surface
        if volumes is None:           # the U-Net's outputs do not exist yet: the same test on what they will be (four channels, one volume per level)
            if not surf.fused_train or not torch.is_tensor(ipts["rays_o"]) or not ipts["rays_o"].is_cuda or torch.cuda.is_current_stream_capturing():
                return False
            from .. import ops
            return (ops.SdfTrainStep.supported(surf.sdf_network, n_feat) and ops.BlendPlan.supported(surf.color_network) and ipts["imgs"].shape[0] >= 2
                    and surf.n_importance > 0)
        return surf._auto_graph_ok(mode, ipts, volumes, list(self.features))

    def forward(self, mode, ipts, cos_anneal_ratio=1.0, step=None):
        """gens.py:124-157.  Training / fine-tune calls run as a CAPTURED step after two eager ones (forward = one HIP graph replay, the backward
        of the caller's loss = a second one; gens_amd.graph.AutoGraph) -- the loop of runner.py:157-166 / 300-308 stays as it is.
        `model.auto_graph = False` or GENS_AUTO_GRAPH=0: every call eager."""
        if not self._auto_graph_ok(mode, ipts, step):
            if getattr(self, "auto_graph", True):
                return self._forward_impl(mode, ipts, cos_anneal_ratio, step)
            self.implicit_surface._auto_suppressed = True       # switched off on the model: off for the render inside it too
            try:
                return self._forward_impl(mode, ipts, cos_anneal_ratio, step)
            finally:
                self.implicit_surface._auto_suppressed = False
        from .. import graph
        surf = self.implicit_surface
        auto = getattr(self, "_auto", None)
        if auto is None:
            auto = self._auto = graph.AutoGraph()
        copied = {k: ipts[k] for k in surf.STEP_INPUTS if k in ipts}
        if self.has_vol:
            copied["view_index"] = self._view_index_of(ipts["view_ids"])
        refs = self._step_refs()
        use_match = not (step is None or step < 5)

        def body(cp, sc, alias):
            ip = dict(ipts)
            ip.update({k: v for k, v in cp.items() if k != "view_index"})
            if "view_index" in cp:
                ip["view_ids"] = cp["view_index"]
            surf._auto_suppressed = True              # the render is part of THIS step's graph (or warm-up), not a captured step of its own
            try:
                return self._forward_impl(mode, ip, sc["cos_anneal_ratio"], 5.0 if use_match else 1.0, reload_match=False)
            finally:
                surf._auto_suppressed = False
        frozen = tuple(f._version for f in self.features) if self.has_vol else ()      # (the cached layouts of the frozen maps belong to these versions)
        key = ("GenS", mode, self.has_vol, use_match, getattr(self, "_mode_version", 0), getattr(self, "_mode_sig", None), self.training, len(refs), frozen)
        return auto.run(key, copied, refs, {"cos_anneal_ratio": float(cos_anneal_ratio)}, body, [surf], module=self)

    def _forward_impl(self, mode, ipts, cos_anneal_ratio=1.0, step=None, reload_match=True):
        if not self.has_vol:
            imgs, intrs, c2ws = ipts["imgs"], ipts["intrs"], ipts["c2ws"]
            features = self.feature_network(imgs)
            if reload_match:
                self._reload_match(step)
            with torch.no_grad():
                match_features = self.match_feature_network(imgs)
            volumes, mask_volmes = self.volume.agg_mean_var(features, intrs, c2ws)
            volumes = self.reg_network(volumes)
        else:
            view_ids = ipts["view_ids"] if mode != "val" else list(range(ipts["imgs"].shape[0]))
            volumes, mask_volmes = list(self.volumes), list(self.mask_volmes)
            index = self._view_index_of(view_ids)
            features = self._select_frozen_views(index)
            match_features = features
        return self.implicit_surface(mode, ipts, volumes, mask_volmes, features, match_features, cos_anneal_ratio, step)
