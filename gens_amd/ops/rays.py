"""K5-K7 (hierarchical sampling), K8 (compositing and the step-boundary kernels around it), K9 (patch reads, surface patch warp), K10 (TV).

Part of gens_amd.ops (see ops/__init__.py); citations are relative to /root/reference."""
from .base import *  # noqa: F401,F403
from .lookup import _mask_args

# ------------------------------------------------------------------------------------------------------------------
# K5-K7  hierarchical sampling
# ------------------------------------------------------------------------------------------------------------------
def upsample(rays_o, rays_d, z, sdf, n_new, masks, inv_s, valid_in=None):
    """up_sample + sample_pdf(det) (implicit_surface.py:60-109): -> z_new (B,n_new), pts_new (B*n_new,3), valid_new bool.
    valid_in (B,n) bool/uint8: mask decisions of the existing samples carried from earlier rounds (None: looked up again)."""
    table, dims, nl, bits = _mask_args(masks)
    b, n = z.shape
    z_new = torch.empty(b, n_new, device=z.device, dtype=_f32)
    pts_new = torch.empty(b * n_new, 3, device=z.device, dtype=_f32)
    valid = torch.empty(b * n_new, device=z.device, dtype=torch.uint8)
    vin = None if valid_in is None else _c(valid_in.reshape(b, n).view(torch.uint8))
    L.call("gens_upsample", L.ptr(_c(rays_o)), L.ptr(_c(rays_d)), L.ptr(_c(z)), L.ptr(_c(sdf)), b, n, n_new, float(inv_s), table,
           dims, nl, bits, L.ptr(vin, torch.uint8), L.ptr(z_new), L.ptr(pts_new), L.ptr(valid, torch.uint8), L.stream(),
           nbytes=b * (8 * n + 17 * n_new + 24 + (n if vin is not None else 0)))
    return z_new, pts_new, valid.view(torch.bool)


def merge_samples(z, z_new, sdf=None, sdf_new=None, valid=None, valid_new=None):
    """cat + sort of cat_z_vals (implicit_surface.py:111-133); the per-sample mask decisions ride along when given.
    -> (z, sdf) or (z, sdf, valid)."""
    b, n = z.shape
    n_new = z_new.shape[1]
    z_out = torch.empty(b, n + n_new, device=z.device, dtype=_f32)
    sdf_out = torch.empty_like(z_out) if sdf is not None else None
    u8 = torch.uint8
    v_in = None if valid is None else _c(valid.reshape(b, n).view(u8))
    v_new = None if valid is None else _c(valid_new.reshape(b, n_new).view(u8))
    v_out = None if valid is None else torch.empty(b, n + n_new, device=z.device, dtype=u8)
    L.call("gens_merge_samples", L.ptr(_c(z)), L.ptr(_c(sdf)) if sdf is not None else None, L.ptr(_c(z_new)),
           L.ptr(_c(sdf_new)) if sdf_new is not None else None, L.ptr(v_in, u8), L.ptr(v_new, u8), b, n, n_new, L.ptr(z_out), L.ptr(sdf_out),
           L.ptr(v_out, u8), L.stream(), nbytes=b * (n + n_new) * ((8 if sdf is None else 16) + (2 if valid is not None else 0)))
    if valid is None:
        return z_out, sdf_out
    return z_out, sdf_out, v_out.view(torch.bool)


def merge_upsample(rays_o, rays_d, z, sdf, valid, z_add, sdf_add, valid_add, n_new, masks, inv_s):
    """One launch per sampling round: merge_samples(z, z_add, sdf, sdf_add, valid, valid_add) followed by upsample(...) of the merged ray
    (implicit_surface.py:111-133 then :60-109).  -> (z, sdf, valid) merged (B, n + n_add) and (z_new (B, n_new), pts_new (B n_new, 3),
    valid_new); bit for bit what the two separate operators return."""
    table, dims, nl, bits = _mask_args(masks)
    b, n = z.shape
    n_add = z_add.shape[1]
    u8 = torch.uint8
    dev = z.device
    z_out = torch.empty(b, n + n_add, device=dev, dtype=_f32)
    sdf_out = torch.empty_like(z_out)
    v_out = torch.empty(b, n + n_add, device=dev, dtype=u8)
    z_new = torch.empty(b, n_new, device=dev, dtype=_f32)
    pts_new = torch.empty(b * n_new, 3, device=dev, dtype=_f32)
    v_new = torch.empty(b * n_new, device=dev, dtype=u8)
    m = n + n_add
    L.call("gens_merge_upsample", L.ptr(_c(rays_o)), L.ptr(_c(rays_d)), L.ptr(_c(z)), L.ptr(_c(sdf)), L.ptr(_c(valid.reshape(b, n).view(u8)), u8),
           L.ptr(_c(z_add)), L.ptr(_c(sdf_add)), L.ptr(_c(valid_add.reshape(b, n_add).view(u8)), u8), b, n, n_add, n_new, float(inv_s), 0.0, table, dims,
           nl, bits, 0, L.ptr(z_out), L.ptr(sdf_out), L.ptr(v_out, u8), L.ptr(z_new), L.ptr(pts_new), L.ptr(v_new, u8), L.stream(),
           nbytes=b * (9 * (n + n_add) + 9 * m + 17 * n_new + 24), label="gens_merge_upsample")
    return z_out, sdf_out, v_out.view(torch.bool), z_new, pts_new, v_new.view(torch.bool)


def merge_mid_points(rays_o, rays_d, z, z_add, masks, sample_dist):
    """The last sampling round and render_core's first step in one launch: merge_samples(z, z_add) (z only, implicit_surface.py:129-131) and
    ray_points(..., mid=True, sample_dist) of the merged ray (:163-173).  -> z (B, n + n_add), pts (B (n + n_add), 3), valid; bit for bit
    what the two separate operators return."""
    table, dims, nl, bits = _mask_args(masks)
    b, n = z.shape
    n_add = z_add.shape[1]
    m = n + n_add
    dev = z.device
    z_out = torch.empty(b, m, device=dev, dtype=_f32)
    pts = torch.empty(b * m, 3, device=dev, dtype=_f32)
    valid = torch.empty(b * m, device=dev, dtype=torch.uint8)
    L.call("gens_merge_upsample", L.ptr(_c(rays_o)), L.ptr(_c(rays_d)), L.ptr(_c(z)), None, None, L.ptr(_c(z_add)), None, None, b, n, n_add, 0, 0.0,
           float(sample_dist), table, dims, nl, bits, 1, L.ptr(z_out), None, None, None, L.ptr(pts), L.ptr(valid, torch.uint8), L.stream(),
           nbytes=b * (4 * (n + n_add) + 17 * m + 24), label="gens_merge_mid_points")
    return z_out, pts, valid.view(torch.bool)


# ------------------------------------------------------------------------------------------------------------------
# K8  compositing (implicit_surface.py:160-168, 202-303)
# ------------------------------------------------------------------------------------------------------------------
def _anneal(cos_anneal):
    """The annealing ratio as the launch takes it: a Python float by value, or a one-element float32 device tensor by address."""
    if torch.is_tensor(cos_anneal):
        assert cos_anneal.is_cuda and cos_anneal.dtype == _f32 and cos_anneal.numel() == 1
        return cos_anneal.detach()
    return float(cos_anneal)


def _composite_in(rays_o, rays_d, z, sdf, grad, color, smooth, voxel_mask, src_vis, inv_s, z_max, sample_dist, cos_anneal, rot):
    ci = L.CompositeIn()
    ci.rays_o, ci.rays_d, ci.z = L.ptr(rays_o), L.ptr(rays_d), L.ptr(z)
    ci.sdf, ci.grad, ci.color, ci.smooth = L.ptr(sdf), L.ptr(grad), L.ptr(color), L.ptr(smooth)
    ci.voxel_mask = L.ptr(voxel_mask, torch.uint8)
    ci.src_vis = L.ptr(src_vis, torch.uint8)
    ci.inv_s, ci.z_max = L.ptr(inv_s), L.ptr(z_max)
    ci.n_rays, ci.n = z.shape
    ci.n_src = src_vis.shape[-1] if src_vis is not None else 0
    ci.sample_dist = float(sample_dist)
    if torch.is_tensor(cos_anneal):          # one float ON THE DEVICE: a captured step is replayed with whatever ratio the caller left there
        ci.cos_anneal, ci.cos_anneal_dev = 0.0, L.ptr(cos_anneal)
    else:
        ci.cos_anneal, ci.cos_anneal_dev = float(cos_anneal), None
    if torch.is_tensor(rot):                 # nine floats on the device (SceneCams.rot_inv): no host read
        ci.rot_dev = L.ptr(rot)
    else:
        ci.rot_dev = None
        for k in range(9):
            ci.rot[k] = rot[k]
    return ci


class _Composite(torch.autograd.Function):
    """inputs with gradient: sdf (B,n), grad (B,n,3), color (B,n,3), smooth (B,n,3)|None, inv_s (1,)"""

    @staticmethod
    def forward(ctx, sdf, grad, color, smooth, inv_s, rays_o, rays_d, z, voxel_mask, src_vis, z_max, sample_dist, cos_anneal, rot):
        b, n = z.shape
        dev = z.device
        sdf, grad, color = _c(sdf.detach()), _c(grad.detach()), _c(color.detach())
        smooth = _c(smooth.detach()) if smooth is not None else None
        inv_s = _c(inv_s.detach().reshape(1))
        ci = _composite_in(rays_o, rays_d, z, sdf, grad, color, smooth, voxel_mask, src_vis, inv_s, z_max, sample_dist, cos_anneal, rot)
        f = lambda *s: torch.empty(*s, device=dev, dtype=_f32)  # noqa: E731
        o = dict(color=f(b, 3), normal=f(b, 3), depth=f(b), wsum=f(b), wmax=f(b), mid_in=f(b), sdf_depth=f(b), z_cross=f(b), eik_num=f(b),
                 eik_den=f(b), smooth_vec=f(b, 3), weights=f(b, n), inside=f(b, n), pts_cross=f(b, 3))
        valid = torch.empty(b, device=dev, dtype=torch.uint8)
        cross_idx = torch.empty(b, device=dev, dtype=torch.int32)
        co = L.CompositeOut()
        for k, t in o.items():
            setattr(co, k, L.ptr(t))
        co.valid = L.ptr(valid, torch.uint8)
        co.cross_idx = L.ptr(cross_idx, torch.int32)
        n_src = src_vis.shape[-1] if src_vis is not None else 0
        L.call("gens_composite_fwd", C.byref(ci), C.byref(co), L.stream(),
               nbytes=b * n * (4 + 4 + 12 + 12 + 1 + n_src + (12 if smooth is not None else 0) + 8) + b * 100)
        ctx.save_for_backward(sdf, grad, color, smooth, inv_s, rays_o, rays_d, z, voxel_mask, src_vis, z_max, o["weights"], cross_idx,
                              o["smooth_vec"])
        ctx.meta = (sample_dist, cos_anneal, rot)
        ctx.set_materialize_grads(False)          # (an output nothing differentiates costs no zero-filled cotangent)
        ctx.mark_non_differentiable(o["wmax"], o["mid_in"], o["eik_den"], o["inside"], valid, cross_idx, o["pts_cross"])
        return (o["color"], o["normal"], o["depth"], o["weights"], o["wsum"], o["eik_num"], o["smooth_vec"], o["z_cross"], o["sdf_depth"],
                o["wmax"], o["mid_in"], o["eik_den"], o["inside"], valid, cross_idx, o["pts_cross"])

    @staticmethod
    def backward(ctx, g_color, g_normal, g_depth, g_weights, g_wsum, g_eik, g_smv, g_zc, _g_sdfdepth, *_unused):
        (sdf, grad, color, smooth, inv_s, rays_o, rays_d, z, voxel_mask, src_vis, z_max, weights, cross_idx, smooth_vec) = ctx.saved_tensors
        sample_dist, cos_anneal, rot = ctx.meta
        b, n = z.shape
        ci = _composite_in(rays_o, rays_d, z, sdf, grad, color, smooth, voxel_mask, src_vis, inv_s, z_max, sample_dist, cos_anneal, rot)
        cg = L.CompositeGrad()
        keep = []

        def cot(t):
            if t is None:
                return None
            t = _c(t.to(_f32))
            keep.append(t)
            return L.ptr(t)
        cg.g_color, cg.g_normal, cg.g_depth, cg.g_weights = cot(g_color), cot(g_normal), cot(g_depth), cot(g_weights)
        cg.g_wsum, cg.g_eik_num, cg.g_smooth_vec, cg.g_z_cross = cot(g_wsum), cot(g_eik), cot(g_smv), cot(g_zc)
        cg.weights, cg.smooth_vec = L.ptr(weights), L.ptr(smooth_vec)
        cg.cross_idx = L.ptr(cross_idx, torch.int32)
        g_sdf = torch.empty_like(sdf)
        g_grad = torch.empty_like(grad)
        g_col = torch.empty_like(color)
        g_smooth = torch.empty_like(smooth) if smooth is not None else None
        g_inv_s = torch.empty(b, device=z.device, dtype=_f32)
        cg.g_sdf, cg.g_grad, cg.g_col, cg.g_smooth, cg.g_inv_s = L.ptr(g_sdf), L.ptr(g_grad), L.ptr(g_col), L.ptr(g_smooth), L.ptr(g_inv_s)
        L.call("gens_composite_bwd", C.byref(ci), C.byref(cg), L.stream())
        return (g_sdf, g_grad, g_col, g_smooth, g_inv_s.sum().reshape(1)) + (None,) * 9


class _InvS(torch.autograd.Function):
    """inv_s = clip(exp(10 variance), 1e-6, 1e6) (variance_network.py:11, implicit_surface.py:206) whose value an earlier launch of the step
    already computed (StepPoints.scalars = [z_max, inv_s, 1 / inv_s, inside the clip range]); only the backward is left to do."""

    @staticmethod
    def forward(ctx, variance, scalars):
        ctx.save_for_backward(scalars)
        return scalars[1:2].clone()

    @staticmethod
    def backward(ctx, g):
        scalars, = ctx.saved_tensors
        return (g * (scalars[1:2] * scalars[3:4] * 10.0)).reshape(()), None


def inv_s_from(variance, scalars):
    return _InvS.apply(variance, scalars)


class _CompositeTrain(torch.autograd.Function):
    """The compositing of a fused TRAINING step: K8 on the first n_ray rows of the step's dense arrays (StepPoints), the two per-batch
    reductions gradient_error / smooth_error (implicit_surface.py:248-253) by one finishing workgroup, inv_s taken from the step's scalars
    with its gradient going straight to `variance` -- two launches forward, two backward, no torch glue.  The gradients of the dense
    arrays come back FULL size (zeros in the rows of the random / pseudo points), so no slice sits in the autograd graph.
    inputs with gradient: y_all (N,1), g_all (N,3), s_all (N,3), color (n_ray,3), variance ()."""

    @staticmethod
    def forward(ctx, y_all, g_all, s_all, color, variance, sel, rays_o, rays_d, z, voxel_mask, src_vis, sample_dist, cos_anneal, rot):
        b, n = z.shape
        dev = z.device
        y_all, g_all, s_all, color = _c(y_all.detach()), _c(g_all.detach()), _c(s_all.detach()), _c(color.detach())
        inv_s, z_max = sel.scalars[1:2], sel.scalars[0:1]
        ci = _composite_in(rays_o, rays_d, z, y_all, g_all, color, s_all, voxel_mask, src_vis, inv_s, z_max, sample_dist, cos_anneal, rot)
        f = lambda *s: torch.empty(*s, device=dev, dtype=_f32)  # noqa: E731
        o = dict(color=f(b, 3), normal=f(b, 3), depth=f(b), wsum=f(b), wmax=f(b), mid_in=f(b), sdf_depth=f(b), z_cross=f(b), eik_num=f(b),
                 eik_den=f(b), smooth_vec=f(b, 3), weights=f(b, n), inside=f(b, n), pts_cross=f(b, 3))
        valid = torch.empty(b, device=dev, dtype=torch.uint8)
        cross_idx = torch.empty(b, device=dev, dtype=torch.int32)
        co = L.CompositeOut()
        for k, t in o.items():
            setattr(co, k, L.ptr(t))
        co.valid = L.ptr(valid, torch.uint8)
        co.cross_idx = L.ptr(cross_idx, torch.int32)
        n_src = src_vis.shape[-1] if src_vis is not None else 0
        L.call("gens_composite_fwd", C.byref(ci), C.byref(co), L.stream(), nbytes=b * n * (4 + 4 + 12 + 12 + 1 + n_src + 12 + 8) + b * 100)
        finish = f(4)
        L.call("gens_composite_finish_fwd", L.ptr(o["eik_num"]), L.ptr(o["eik_den"]), L.ptr(o["smooth_vec"]), b, L.ptr(finish), L.stream())
        ctx.save_for_backward(y_all, g_all, s_all, color, rays_o, rays_d, z, voxel_mask, src_vis, sel.scalars, o["weights"], cross_idx, o["smooth_vec"],
                              finish)
        ctx.meta = (sample_dist, cos_anneal, rot)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(o["wmax"], o["mid_in"], o["inside"], valid, cross_idx, o["pts_cross"])
        return (o["color"], o["normal"], o["depth"], o["weights"], o["wsum"], o["z_cross"], o["sdf_depth"], finish[2], finish[3], o["wmax"], o["mid_in"],
                o["inside"], valid, cross_idx, o["pts_cross"])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_color, g_normal, g_depth, g_weights, g_wsum, g_zc, _g_sdfdepth, g_ge, g_se, *_unused):
        (y_all, g_all, s_all, color, rays_o, rays_d, z, voxel_mask, src_vis, scalars, weights, cross_idx, smooth_vec, finish) = ctx.saved_tensors
        sample_dist, cos_anneal, rot = ctx.meta
        b, n = z.shape
        n_all = y_all.shape[0]
        ci = _composite_in(rays_o, rays_d, z, y_all, g_all, color, s_all, voxel_mask, src_vis, scalars[1:2], scalars[0:1], sample_dist, cos_anneal, rot)
        cg = L.CompositeGrad()
        keep = []

        def cot(t):
            if t is None:
                return None
            t = _c(t.to(_f32))
            keep.append(t)
            return L.ptr(t)
        cg.g_color, cg.g_normal, cg.g_depth, cg.g_weights = cot(g_color), cot(g_normal), cot(g_depth), cot(g_weights)
        cg.g_wsum, cg.g_eik_num, cg.g_smooth_vec, cg.g_z_cross = cot(g_wsum), None, None, cot(g_zc)
        cg.g_gradient_error, cg.g_smooth_error, cg.finish = cot(None if g_ge is None else g_ge.reshape(1)), cot(None if g_se is None else g_se.reshape(1)), L.ptr(finish)
        cg.weights, cg.smooth_vec = L.ptr(weights), L.ptr(smooth_vec)
        cg.cross_idx = L.ptr(cross_idx, torch.int32)
        dev = z.device
        g_sdf = torch.empty(n_all, 1, device=dev, dtype=_f32)
        g_grad = torch.empty(n_all, 3, device=dev, dtype=_f32)
        g_smooth = torch.empty(n_all, 3, device=dev, dtype=_f32)
        g_col = torch.empty_like(color)
        g_inv_s = torch.empty(b, device=dev, dtype=_f32)
        g_var = torch.empty((), device=dev, dtype=_f32)
        cg.g_sdf, cg.g_grad, cg.g_col, cg.g_smooth, cg.g_inv_s = L.ptr(g_sdf), L.ptr(g_grad), L.ptr(g_col), L.ptr(g_smooth), L.ptr(g_inv_s)
        L.call("gens_composite_bwd", C.byref(ci), C.byref(cg), L.stream())
        L.call("gens_composite_finish_bwd", L.ptr(g_inv_s), b, L.ptr(scalars), L.ptr(g_var), L.ptr(g_sdf), L.ptr(g_grad), L.ptr(g_smooth), b * n, n_all,
               L.stream())
        return (g_sdf, g_grad, g_smooth, g_col, g_var) + (None,) * 9


COMPOSITE_TRAIN_KEYS = ("color", "normal", "depth", "weights", "wsum", "z_cross", "sdf_depth", "gradient_error", "smooth_error", "wmax", "mid_in",
                        "inside", "valid", "cross_idx", "pts_cross")


def composite_train(sel, rays_o, rays_d, z, sample_dist, y_all, g_all, s_all, color, variance, voxel_mask, src_vis, cos_anneal, rot):
    """-> dict keyed by COMPOSITE_TRAIN_KEYS.  sel: the step's ops.StepPoints; voxel_mask (n_ray,) uint8 / bool, src_vis (n_ray, S) uint8 / bool."""
    b, n = z.shape
    u8 = torch.uint8
    vm = voxel_mask.reshape(b * n)
    vm = _c(vm.view(u8) if vm.dtype == torch.bool else vm.to(u8))
    sv = src_vis.reshape(b * n, -1)
    sv = _c(sv.view(u8) if sv.dtype == torch.bool else sv.to(u8))
    outs = _CompositeTrain.apply(y_all, g_all, s_all, color, variance, sel, _c(rays_o.to(_f32)), _c(rays_d.to(_f32)), _c(z.detach().to(_f32)), vm, sv,
                                 float(sample_dist), _anneal(cos_anneal), rot)
    return dict(zip(COMPOSITE_TRAIN_KEYS, outs))


COMPOSITE_KEYS = ("color", "normal", "depth", "weights", "wsum", "eik_num", "smooth_vec", "z_cross", "sdf_depth", "wmax", "mid_in", "eik_den",
                  "inside", "valid", "cross_idx", "pts_cross")


def composite(rays_o, rays_d, z, sample_dist, sdf, gradients, smooth, color, voxel_mask, src_vis, inv_s, cos_anneal, c2w_ref, z_max=None):
    """Everything render_core computes after the networks have run; returns a dict keyed by COMPOSITE_KEYS.
    z_max: optional (1,) device tensor holding max(z) (implicit_surface.py:301) when an earlier launch already reduced it."""
    b, n = z.shape
    # R_ref^-1 (implicit_surface.py:242,245) travels by value in the launch block; a list from Scene.ref_rotation() avoids the
    # device->host read (a synchronisation) on every ray chunk
    if isinstance(c2w_ref, (list, tuple)) or (torch.is_tensor(c2w_ref) and c2w_ref.numel() == 9):
        rot = c2w_ref                        # host floats, or SceneCams.rot_inv on the device
    else:
        rot = _c(inv(c2w_ref[:3, :3].to(_f32)).reshape(-1))
    z = _c(z.detach().to(_f32))
    if z_max is None:
        z_max = z.max().reshape(1)                                              # implicit_surface.py:301
    u8 = torch.uint8
    vm = voxel_mask.reshape(b * n)
    vm = _c(vm.view(u8) if vm.dtype == torch.bool else vm.to(u8))
    sv = None
    if src_vis is not None:
        sv = src_vis.reshape(b * n, -1)
        sv = _c(sv.view(u8) if sv.dtype == torch.bool else sv.to(u8))
    outs = _Composite.apply(sdf.reshape(b, n), gradients.reshape(b, n, 3), color.reshape(b, n, 3),
                            smooth.reshape(b, n, 3) if smooth is not None else None, inv_s.reshape(1), _c(rays_o.to(_f32)),
                            _c(rays_d.to(_f32)), z, vm, sv, z_max, float(sample_dist), _anneal(cos_anneal), rot)
    return dict(zip(COMPOSITE_KEYS, outs))


# ------------------------------------------------------------------------------------------------------------------
# K9  patch reads (projector.py:406-416) and the feature up-sampling that feeds them (implicit_surface.py:313-326)
# ------------------------------------------------------------------------------------------------------------------
class _PatchSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xy, image_tex, c):
        h, w, _ = image_tex.shape
        xy_c = _c(xy.detach().to(_f32))
        p = xy_c.shape[0]
        out = torch.empty(p, c, device=xy.device, dtype=_f32)
        L.call("gens_patch_sample_fwd", L.ptr(image_tex), h, w, c, L.ptr(xy_c), p, L.ptr(out), L.stream())
        ctx.save_for_backward(xy_c, image_tex)
        ctx.c = c
        return out

    @staticmethod
    def backward(ctx, g_out):
        xy, image_tex = ctx.saved_tensors
        h, w, _ = image_tex.shape
        g_xy = torch.empty_like(xy)
        L.call("gens_patch_sample_bwd", L.ptr(image_tex), h, w, ctx.c, L.ptr(xy), L.ptr(_c(g_out)), xy.shape[0], L.ptr(g_xy), L.stream())
        return g_xy, None, None


def patch_sample(image_tex, xy, channels):
    """image_tex (H,W,C_pad) texels (constant), xy (P,2) pixel coordinates -> (P,C); differentiable in xy."""
    return _PatchSample.apply(xy, image_tex, channels)


class _PatchWarp(torch.autograd.Function):
    """surface_patch_warp (projector.py:353-437) fused: (z_cross (B), rays_o, rays_d, g0 (B,3), cams, texels) -> (ref (1,B,P,C), sampled
    (S,B,P,C)); differentiable with respect to z_cross (the normal is used detached, implicit_surface.py:306-310)."""

    @staticmethod
    def forward(ctx, z, rays_o, rays_d, g0, cams, tex, c, patch):
        nv, h, w, _ = tex.shape
        b = z.shape[0]
        dev = z.device
        z_c, o_c, d_c, g_c = _c(z.detach().to(_f32)), _c(rays_o.detach().to(_f32)), _c(rays_d.detach().to(_f32)), _c(g0.detach().to(_f32).reshape(b, 3))
        p = patch * patch
        ref = torch.empty(1, b, p, c, device=dev, dtype=_f32)
        sampled = torch.empty(nv - 1, b, p, c, device=dev, dtype=_f32)
        ctx.args = (L.ptr(o_c), L.ptr(d_c), L.ptr(z_c), L.ptr(g_c), b, L.ptr(cams.c2w), L.ptr(cams.intr), L.ptr(cams.kinv_ref), nv, L.ptr(tex, align=16),
                    h, w, c, patch)
        ctx.keep = (o_c, d_c, z_c, g_c, cams, tex)
        L.call("gens_patch_warp_fwd", *ctx.args, L.ptr(ref), L.ptr(sampled), L.stream(), nbytes=4 * nv * b * p * c)
        ctx.mark_non_differentiable(ref)
        return ref, sampled

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, _g_ref, g_sampled):
        b = ctx.args[4]
        g_z = torch.empty(b, device=g_sampled.device, dtype=_f32)
        L.call("gens_patch_warp_bwd", *ctx.args, L.ptr(_c(g_sampled.to(_f32))), L.ptr(g_z), L.stream(), nbytes=8 * (ctx.args[8] - 1) * b * ctx.args[13] ** 2 * ctx.args[12])
        return g_z, None, None, None, None, None, None, None


def patch_warp(z_cross, rays_o, rays_d, g0, cams, warp, patch_size=11):
    """warp: the (texels (nv,H,W,C_pad), C) pair of build_warp_features."""
    tex, c = warp
    return _PatchWarp.apply(z_cross, rays_o, rays_d, g0, cams, aligned16(tex), int(c), int(patch_size))


def build_warp_features(levels):
    """cat([f0, up(f1), up(f2)], 1) of implicit_surface.py:313-326 as (nv,H,W,12) texels; inputs (nv,4,h_i,w_i) NCHW, detached.
    The result is kept on levels[0] for the current versions of the three maps (frozen maps -- fine-tuning -- are up-sampled once)."""
    key = tuple((id(f), f._version) for f in levels)
    hit = getattr(levels[0], "_gens_warp", None)
    if hit is not None and hit[0] == key and all(a is b for a, b in zip(hit[1], levels[1:])):
        return hit[2]
    res = _build_warp_features(levels)
    try:
        levels[0]._gens_warp = (key, list(levels[1:]), res)        # (the coarser maps are held so that their ids cannot be recycled)
    except (AttributeError, RuntimeError):
        pass
    return res


def _build_warp_features(levels):
    f0 = levels[0].detach()
    nv, c, h, w = f0.shape
    ctot = sum(f.shape[1] for f in levels)
    cpad = 4 * ((ctot + 3) // 4)
    maps = [_c(f.detach().to(_f32)) for f in levels]
    if len(maps) <= 8:                      # one launch, every texel written whole (gens_upsample2d_cat)
        dst = torch.empty(nv, h, w, cpad, device=f0.device, dtype=_f32)
        chw = [d for f in maps for d in f.shape[1:]]
        L.call("gens_upsample2d_cat", L.ptr_table(maps), L.int_table(chw), len(maps), nv, L.ptr(dst), h, w, cpad, L.stream())
        return dst, ctot
    dst = torch.zeros(nv, h, w, cpad, device=f0.device, dtype=_f32)
    off = 0
    for f in maps:
        L.call("gens_upsample2d_into", L.ptr(f), nv, f.shape[1], f.shape[2], f.shape[3], L.ptr(dst), h, w, cpad, off, L.stream())
        off += f.shape[1]
    return dst, ctot


# ------------------------------------------------------------------------------------------------------------------
# K10  tv_regularization (implicit_surface.py:135-150)
# ------------------------------------------------------------------------------------------------------------------
class _TVLevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol, mask):
        _, c, x, y, z = vol.shape
        assert c == 4
        vol_c, mask_c = _c(vol.detach()), _c(mask.detach())
        nb = L.load().gens_tv_blocks(x * y * z)
        partial = torch.empty(nb, 4, device=vol.device, dtype=_f32)
        L.call("gens_tv_fwd", L.ptr(vol_c), L.ptr(mask_c), x, y, z, L.ptr(partial), L.stream())
        sums = partial.double().sum(0)
        den = sums[3] + 1e-8                                                   # (Q13): all three axes use mx's count
        tv = torch.sqrt((sums[0] + sums[1] + sums[2]) / den).to(_f32)
        ctx.save_for_backward(vol_c, mask_c, tv, den.to(_f32))
        return tv

    @staticmethod
    def backward(ctx, g):
        vol, mask, tv, den = ctx.saved_tensors
        _, _, x, y, z = vol.shape
        coef = _c((g / (2.0 * tv * den)).to(_f32).reshape(1))                  # stays on the device (float(...) here stalled the host once per level)
        g_vol = torch.empty_like(vol)
        L.call("gens_tv_bwd_scaled", L.ptr(vol), L.ptr(mask), x, y, z, 1.0, L.ptr(coef), L.ptr(g_vol), L.stream())
        return g_vol, None


class _TVLevels(torch.autograd.Function):
    """tv_regularization of all levels: two launches forward (partial sums, one finishing workgroup), one backward."""

    @staticmethod
    def forward(ctx, n, *vm):
        vols, masks = [_c(v.detach()) for v in vm[:n]], [_c(m.detach()) for m in vm[n:]]
        dims = [d for v in vols for d in v.shape[-3:]]
        dev = vols[0].device
        partial = torch.empty(L.load().gens_tv_levels_blocks(L.int_table(dims), n), 4, device=dev, dtype=_f32)
        out = torch.empty(1 + n, device=dev, dtype=_f32)
        L.call("gens_tv_levels_fwd", L.ptr_table(vols, align=16), L.ptr_table(masks, align=16), L.int_table(dims), n, L.ptr(partial), L.ptr(out), L.stream(),
               nbytes=sum(20 * v[0, 0].numel() for v in vols))
        ctx.save_for_backward(out, *vols, *masks)
        ctx.n, ctx.dims = n, dims
        return out[0]

    @staticmethod
    def backward(ctx, g):
        out, *vm = ctx.saved_tensors
        n = ctx.n
        vols, masks = vm[:n], vm[n:]
        g_vols = [torch.empty_like(v) for v in vols]
        L.call("gens_tv_levels_bwd", L.ptr_table(list(vols), align=16), L.ptr_table(list(masks), align=16), L.int_table(ctx.dims), n, L.ptr(out),
               L.ptr(_c(g.detach().to(_f32).reshape(1))), L.ptr_table(g_vols, align=16), L.stream(), nbytes=sum(36 * v[0, 0].numel() for v in vols))
        return (None, *g_vols, *([None] * n))


def tv_levels_ok(volumes, masks):
    """The fused all-level kernels cover 4-channel levels with Z % 4 == 0 below 2^31 voxels on 16-byte aligned storage."""
    return all(v.dim() == 5 and v.shape[1] == 4 and v.shape[-1] % 4 == 0 and v[0, 0].numel() < 2 ** 31 and v.is_contiguous() and v.data_ptr() % 16 == 0
               and m.is_contiguous() and m.data_ptr() % 16 == 0 for v, m in zip(volumes, masks)) and len(volumes) <= L.MAX_LEVELS


def tv_regularization(volumes, masks):
    volumes, masks = list(volumes), list(masks)
    if volumes and volumes[0].is_cuda and tv_levels_ok(volumes, masks):
        return _TVLevels.apply(len(volumes), *volumes, *masks)
    total = 0
    for lvl, (v, m) in enumerate(zip(volumes, masks)):
        total = total + _TVLevel.apply(v, m) * 0.5 ** lvl
    return total


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]      # private helpers travel too: the package namespace is the old module's
