"""Mirror of the reference's models/losses/loss.py: same class name, configuration keys, forward signature and result keys.

On the device the ~40 element-wise / reduction launches of Loss.forward and the ~45 of its backward are ONE launch each (gens_loss_fwd /
gens_loss_bwd); the patch statistic is K13 (compute_LNCC).  Tensors that do not live on the GPU take the reference's own expressions."""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import lib as L
from .ncc import compute_LNCC

_f32 = torch.float32
LOSS_KEYS = ("loss", "color_loss", "eikonal_loss", "sparse_loss", "mfc_loss", "smooth_loss", "tv_loss", "depth_loss", "pseudo_sdf_loss", "pseudo_depth_loss")


def _flat(t):
    t = t.detach().to(_f32).reshape(-1)
    return t if t.is_contiguous() else t.contiguous()


class _FusedLoss(torch.autograd.Function):
    """inputs with gradient: color (B,3), sparse (N,1), pseudo (P,1)|None, ncc (B,1), depth (B,), gradient_error, smooth_error, tv_reg."""

    @staticmethod
    def forward(ctx, color, sparse, pseudo, ncc, depth, ge, se, tv, target, valid, mid_in, pseudo_depth_t, depth_t, weights, sparse_scale):
        dev = color.device
        a = L.LossArgs()
        keep = {}

        def dp(name, t, dtype=_f32):
            if t is None:
                setattr(a, name, None)
                return None
            t = _flat(t) if dtype == _f32 else t.detach().reshape(-1).contiguous()
            keep[name] = t
            setattr(a, name, L.ptr(t, dtype))
            return t
        c = dp("color", color)
        dp("target", target)
        v = valid.detach().reshape(-1)
        v = v.view(torch.uint8) if v.dtype == torch.bool else v.to(torch.uint8)
        dp("valid", v, torch.uint8)
        a.b = c.numel() // 3
        sp = dp("sparse", sparse)
        a.n_sparse = sp.numel()
        a.sparse_scale = float(sparse_scale)
        ps = dp("pseudo", pseudo)
        a.n_pseudo = 0 if ps is None else ps.numel()
        dp("ncc", ncc)
        dp("mid_in", mid_in)
        dp("depth", depth)
        dp("pseudo_depth_t", pseudo_depth_t)
        dp("depth_t", depth_t)
        dp("ge", ge)
        dp("se", se)
        dp("tv", tv)
        for k, w in zip(("w_color", "w_igr", "w_sparse", "w_mfc", "w_smooth", "w_tv", "w_pseudo_sdf", "w_pseudo_depth"), weights):
            setattr(a, k, float(w))
        out = torch.empty(16, device=dev, dtype=_f32)
        a.out = L.ptr(out)
        a.g = a.g_color = a.g_sparse = a.g_pseudo = a.g_ncc = a.g_depth = a.g_scalars = None
        L.call("gens_loss_fwd", C.byref(a), L.stream())
        ctx.args, ctx.keep, ctx.out = a, keep, out
        ctx.shapes = (color.shape, sparse.shape, None if pseudo is None else pseudo.shape, ncc.shape, None if depth is None else depth.shape)
        ctx.set_materialize_grads(False)
        terms = out[:10]
        ctx.mark_non_differentiable(terms)
        return out[0], terms

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, _g_terms):
        a, keep = ctx.args, ctx.keep
        dev = ctx.out.device
        c_shape, s_shape, p_shape, n_shape, d_shape = ctx.shapes
        if g is None:
            return (None,) * 15
        gg = g.detach().to(_f32).reshape(1).contiguous()
        a.g = L.ptr(gg)
        need = ctx.needs_input_grad
        e = lambda shape: torch.empty(shape, device=dev, dtype=_f32)  # noqa: E731
        g_color = e(c_shape) if need[0] else None
        g_sparse = e(s_shape) if need[1] else None
        g_pseudo = e(p_shape) if (need[2] and p_shape is not None) else None
        g_ncc = e(n_shape) if need[3] else None
        g_depth = e(d_shape) if (need[4] and d_shape is not None and "pseudo_depth_t" in keep) else None
        g_sc = e(3)
        a.g_color, a.g_sparse, a.g_pseudo, a.g_ncc, a.g_depth, a.g_scalars = (L.ptr(g_color), L.ptr(g_sparse), L.ptr(g_pseudo), L.ptr(g_ncc),
                                                                             L.ptr(g_depth), L.ptr(g_sc))
        L.call("gens_loss_bwd", C.byref(a), L.stream())
        return (g_color, g_sparse, g_pseudo, g_ncc, g_depth, g_sc[0] if need[5] else None, g_sc[1] if need[6] else None, g_sc[2] if need[7] else None,
                None, None, None, None, None, None, None)


class Loss(nn.Module):
    def __init__(self, confs):
        super().__init__()
        self.color_weight = confs.get_float("color_weight")
        self.sparse_scale_factor = confs.get_float("sparse_scale_factor")
        self.sparse_weight = confs.get_float("sparse_weight")
        self.igr_weight = confs.get_float("igr_weight")
        self.mfc_weight = confs.get_float("mfc_weight")
        self.smooth_weight = confs.get_float("smooth_weight")
        self.tv_weight = confs.get_float("tv_weight")
        self.depth_weight = confs.get_float("depth_weight", default=0.0)
        self.pseudo_sdf_weight = confs.get_float("pseudo_sdf_weight", default=0.0)
        self.pseudo_depth_weight = confs.get_float("pseudo_depth_weight", default=0.0)
        self.fused = True

    def _weights(self):
        return (self.color_weight, self.igr_weight, self.sparse_weight, self.mfc_weight, self.smooth_weight, self.tv_weight, self.pseudo_sdf_weight,
                self.pseudo_depth_weight)

    def forward(self, preds, targets, step=None):
        color = preds["color_fine"]
        scalars = [preds["gradient_error"], preds["smooth_error"], preds["tv_reg"]]
        if self.fused and color.is_cuda and all(t.numel() == 1 for t in scalars):
            ncc = compute_LNCC(preds["ref_gray_val"], preds["sampled_gray_val"])
            loss, terms = _FusedLoss.apply(color, preds["sparse_sdf"], preds.get("pseudo_sdf"), ncc, preds.get("render_depth"), *scalars,
                                           targets["color"], preds["valid_mask"], preds["mid_inside_sphere"], targets.get("pseudo_depth"),
                                           targets.get("depth"), self._weights(), self.sparse_scale_factor)
            out = {k: terms[i] for i, k in enumerate(LOSS_KEYS)}
            out["loss"] = loss
            return out
        return self._forward_torch(preds, targets)

    def _forward_torch(self, preds, targets):
        """The reference's expressions (loss.py:24-93), for tensors that are not on the GPU."""
        valid_mask = preds["valid_mask"]
        color_loss = F.l1_loss(preds["color_fine"], targets["color"], reduction="none")
        color_loss = (color_loss * valid_mask.float()).sum() / (valid_mask.float().sum() + 1e-5)
        eikonal_loss = preds["gradient_error"].mean()
        sparse_loss = torch.exp(-torch.abs(preds["sparse_sdf"]) * self.sparse_scale_factor).mean()
        smooth_loss = preds["smooth_error"].mean()
        tv_loss = preds["tv_reg"].mean()
        ncc = compute_LNCC(preds["ref_gray_val"], preds["sampled_gray_val"])
        ncc_mask = valid_mask * preds["mid_inside_sphere"]
        mfc_loss = 0.5 * ((ncc * ncc_mask).sum(dim=0) / (ncc_mask.sum(dim=0) + 1e-8)).squeeze(-1)
        zero = torch.tensor(0.0).type_as(mfc_loss)
        pseudo_sdf_loss = torch.abs(preds["pseudo_sdf"]).mean() if "pseudo_sdf" in preds else zero

        def masked_l1(t):
            on = (t > 0).float()
            return ((preds["render_depth"] - t).abs() * on).sum() / (on.sum() + 1e-8)
        pseudo_depth_loss = masked_l1(targets["pseudo_depth"]) if "pseudo_depth" in targets else zero
        depth_loss = masked_l1(targets["depth"]) if "depth" in targets else zero
        loss = (color_loss * self.color_weight + eikonal_loss * self.igr_weight + sparse_loss * self.sparse_weight + mfc_loss * self.mfc_weight
                + smooth_loss * self.smooth_weight + tv_loss * self.tv_weight + pseudo_sdf_loss * self.pseudo_sdf_weight
                + pseudo_depth_loss * self.pseudo_depth_weight)
        return {"loss": loss, "color_loss": color_loss, "eikonal_loss": eikonal_loss, "sparse_loss": sparse_loss, "mfc_loss": mfc_loss,
                "smooth_loss": smooth_loss, "tv_loss": tv_loss, "depth_loss": depth_loss, "pseudo_sdf_loss": pseudo_sdf_loss,
                "pseudo_depth_loss": pseudo_depth_loss}
