"""Mirror of the reference's models/modules/volume.py: `Volume.agg_mean_var` (:13-63) on the K1 HIP kernel."""
import torch.nn as nn

from ... import ops


class Volume(nn.Module):
    def __init__(self, confs):
        super().__init__()
        self.volume_dims = confs.get_list("volume_dims")

    def agg_mean_var(self, features, intrs, c2ws, min_vis_view=1):
        """features[i] (nv,4,H/2^i,W/2^i) pairs with volume_dims[i] and intrinsics rows 0-1 * 0.5^i (Q2).

        -> volumes [(1,8,D,D,D) = mean(4)|var(4)], mask_volumes [(1,1,D,D,D) = (#visible views > min_vis_view)].
        Differentiable w.r.t. the features (the projection grid is constant, volume.py:27-44).
        """
        return ops.volume_build(features[:len(self.volume_dims)], intrs, c2ws, self.volume_dims, min_vis_view)
