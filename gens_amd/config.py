"""A minimal stand-in for pyhocon's ConfigTree (not installed here): dotted keys + get_int/get_float/get_list/get_bool.
The model code accepts either this or a real ConfigTree ("Config / flags")."""


class Conf(dict):
    def _walk(self, key):
        node = self
        for part in key.split("."):
            node = dict.__getitem__(node, part)
        return node

    def __getitem__(self, key):
        v = self._walk(key)
        return Conf(v) if isinstance(v, dict) and not isinstance(v, Conf) else v

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    get_int = get_float = get_list = get_bool = get_string = get


def gens_loss_conf(finetune=False):
    """The `train.loss` block of confs/gens.conf:47-59 (finetune: confs/gens_finetune.conf:32-41)."""
    if finetune:
        return Conf(dict(color_weight=1.0, sparse_weight=0.0, igr_weight=0.1, sparse_scale_factor=100, mfc_weight=1.0, smooth_weight=0.0005,
                         tv_weight=0.0001, pseudo_sdf_weight=1.0))
    return Conf(dict(color_weight=1.0, sparse_weight=0.02, igr_weight=0.1, sparse_scale_factor=100, mfc_weight=1.0, smooth_weight=0.0001,
                     tv_weight=0.0001, depth_weight=0.0, pseudo_sdf_weight=1.0, normal_weight=0.0, pseudo_depth_weight=0.05))


def gens_model_conf(volume_dims=(256, 128, 64, 32, 16), n_feature_levels=5, has_vol=False):
    """The `model` block of confs/gens.conf:59-99 for `len(volume_dims)` volume scales."""
    n = len(volume_dims)
    return Conf({
        "has_vol": has_vol,
        "feature_network": {"d_out": [4] * n_feature_levels},
        "volume": {"volume_dims": list(volume_dims)},
        "reg_network": {"d_voluem": [8] * n, "d_out": [4] * n, "d_base": 8},
        "implicit_surface": {
            "sdf_network": dict(d_out=129, d_in=3, d_hidden=128, n_layers=6, skip_in=[3], multires=4, bias=0.5, scale=1.0,
                                geometric_init=True, weight_norm=True, feat_channels=4 * n),
            "color_network": dict(d_feature=4 * n_feature_levels),
            "variance_network": dict(init_val=0.3),
            "render": dict(n_samples=64, n_importance=64, up_sample_steps=4, perturb=1.0),
        },
    })
