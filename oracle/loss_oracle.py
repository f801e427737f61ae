"""CPU oracle of the reference's training loss -- TEST INFRASTRUCTURE, never the product path (see gens_oracle.py's header).

Restates models/losses/loss.py:23-84 term by term in plain torch on the CPU, with the patch statistic from gens_oracle.lncc
(models/losses/ncc.py:7-55).  Pinned by golden g19 (the reference's own Loss.forward on stored predictions, every returned term and the
gradient of `loss` with respect to every differentiable prediction; tests/test_oracle_golden.py).  Citations are relative to /root/reference.
"""
import torch

from . import gens_oracle as K

TERMS = ("loss", "color_loss", "eikonal_loss", "sparse_loss", "mfc_loss", "smooth_loss", "tv_loss", "depth_loss", "pseudo_sdf_loss", "pseudo_depth_loss")


def _masked_l1(pred, target):
    """loss.py:45-46, 50-51: |pred - target| over the positive targets."""
    on = (target > 0).float()
    return ((pred - target).abs() * on).sum() / (on.sum() + 1e-8)


def loss(preds, targets, weights):
    """weights: dict with the configuration keys of loss.py:12-21 (color_weight, sparse_scale_factor, sparse_weight, igr_weight, mfc_weight,
    smooth_weight, tv_weight, pseudo_sdf_weight, pseudo_depth_weight; the last two default to 0 like confs.get_float(..., default=0.0)).
    -> the dict loss.py:71-82 returns."""
    valid = preds["valid_mask"].float()
    color = (preds["color_fine"] - targets["color"]).abs()                                     # F.l1_loss(reduction='none'), loss.py:26
    color_loss = (color * valid).sum() / (valid.sum() + 1e-5)                                  # :27
    eikonal_loss = preds["gradient_error"].mean()                                              # :29
    sparse_loss = torch.exp(-preds["sparse_sdf"].abs() * weights["sparse_scale_factor"]).mean()    # :31
    smooth_loss = preds["smooth_error"].mean()                                                 # :33
    tv_loss = preds["tv_reg"].mean()                                                           # :35
    ncc = K.lncc(preds["ref_gray_val"], preds["sampled_gray_val"])                             # :37
    ncc_mask = valid * preds["mid_inside_sphere"]                                              # :38
    mfc_loss = 0.5 * ((ncc * ncc_mask).sum(dim=0) / (ncc_mask.sum(dim=0) + 1e-8)).squeeze(-1)  # :39
    zero = torch.tensor(0.0).type_as(mfc_loss)
    pseudo_sdf_loss = preds["pseudo_sdf"].abs().mean() if "pseudo_sdf" in preds else zero      # :41-44
    pseudo_depth_loss = _masked_l1(preds["render_depth"], targets["pseudo_depth"]) if "pseudo_depth" in targets else zero   # :46-49
    depth_loss = _masked_l1(preds["render_depth"], targets["depth"]) if "depth" in targets else zero                       # :51-54
    total = (color_loss * weights["color_weight"] + eikonal_loss * weights["igr_weight"] + sparse_loss * weights["sparse_weight"]
             + mfc_loss * weights["mfc_weight"] + smooth_loss * weights["smooth_weight"] + tv_loss * weights["tv_weight"]
             + pseudo_sdf_loss * weights.get("pseudo_sdf_weight", 0.0) + pseudo_depth_loss * weights.get("pseudo_depth_weight", 0.0))   # :63-70
    return {"loss": total, "color_loss": color_loss, "eikonal_loss": eikonal_loss, "sparse_loss": sparse_loss, "mfc_loss": mfc_loss,
            "smooth_loss": smooth_loss, "tv_loss": tv_loss, "depth_loss": depth_loss, "pseudo_sdf_loss": pseudo_sdf_loss,
            "pseudo_depth_loss": pseudo_depth_loss}
