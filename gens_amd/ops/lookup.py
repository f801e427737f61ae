"""K2 / K2'' (lookup_volume with first and second derivatives), K3 (nearest masks, ray points), K4 (lookup_feature).

Part of gens_amd.ops (see ops/__init__.py); citations are relative to /root/reference."""
from .base import *  # noqa: F401,F403

# ------------------------------------------------------------------------------------------------------------------
# K2 / K2''  lookup_volume(..., "grad") with first and second derivatives
# ------------------------------------------------------------------------------------------------------------------
def _vset(layout, vols):
    return VolumeSet(list(vols), layout)


class _Lookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, layout, *vols):
        vs = _vset(layout, [v.detach() for v in vols])
        pts_c = _c(pts.detach().to(_f32))
        n = pts_c.shape[0]
        out = torch.empty(n, 4 * vs.n, device=pts.device, dtype=_f32)
        L.call("gens_lookup_volume_fwd", vs.table, vs.dim_table, vs.n, layout, L.ptr(pts_c), n, L.ptr(out), L.stream(),
               nbytes=n * (12 + 16 * vs.n))
        ctx.save_for_backward(pts, *vols)   # the INPUT tensors: the second backward must reach their producers
        ctx.layout = layout
        return out

    @staticmethod
    def backward(ctx, g_out):
        pts, *vols = ctx.saved_tensors
        want_vol = any(ctx.needs_input_grad[2:])
        res = _LookupBwd.apply(g_out, pts, ctx.layout, want_vol, *vols)
        return (res[0], None) + tuple(res[1:])


def _bricks_scratch(n, device):
    return torch.empty(L.load().gens_lookup_scatter_bricks_scratch_bytes(n), device=device, dtype=torch.uint8)


class _LookupBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g_out, pts, layout, want_vol, *vols):
        vs = _vset(layout, [v.detach() for v in vols])
        n = pts.shape[0]
        g_out_c = _c(g_out.detach().to(_f32))
        pts_c = _c(pts.detach().to(_f32))
        g_pts = torch.empty(n, 3, device=pts.device, dtype=_f32)
        g_vols = [torch.zeros_like(v) for v in vs.tensors] if want_vol else None
        if want_vol and n >= kernels.k2_bricks_min:
            # large point sets: the volume-gradient scatter brick by brick (every touched voxel written once), not 32 atomics per point and level
            scratch = _bricks_scratch(n, pts.device)
            L.call("gens_lookup_volume_bwd_bricks", vs.table, vs.dim_table, vs.n, layout, L.ptr(pts_c), L.ptr(g_out_c), n, L.ptr_table(g_vols),
                   L.ptr(g_pts), L.ptr(scratch, torch.uint8), scratch.numel(), L.stream(), nbytes=n * (24 + 16 * vs.n), label="gens_lookup_volume_bwd")
        else:
            L.call("gens_lookup_volume_bwd", vs.table, vs.dim_table, vs.n, layout, L.ptr(pts_c), L.ptr(g_out_c), n, L.ptr_table(g_vols),
                   L.ptr(g_pts), L.stream(), nbytes=n * (24 + 16 * vs.n))
        ctx.save_for_backward(g_out_c, pts_c, *vols)
        ctx.layout, ctx.want_vol = layout, want_vol
        if want_vol:
            return (g_pts,) + tuple(g.reshape(v.shape) for g, v in zip(g_vols, vols))
        return (g_pts,) + tuple(None for _ in vols)

    @staticmethod
    def backward(ctx, gg_pts, *gg_vols):
        g_out, pts, *vols = ctx.saved_tensors
        layout = ctx.layout
        vs = _vset(layout, [v.detach() for v in vols])
        n = pts.shape[0]
        if gg_pts is None:
            gg_pts = torch.zeros(n, 3, device=pts.device, dtype=_f32)
        have_ggv = any(g is not None for g in gg_vols)
        ggv = None
        if have_ggv:  # cuda_gridsample.py:113-114 allocates zeros here; missing levels are simply NULL-safe zeros
            ggv = [_c(g.detach().reshape(t.shape)) if g is not None else torch.zeros_like(t) for g, t in zip(gg_vols, vs.tensors)]
        want_vol = any(ctx.needs_input_grad[4:])
        g_vols2 = [torch.zeros_like(t) for t in vs.tensors] if want_vol else None
        gg_out = torch.empty_like(g_out)
        g_pts2 = torch.empty(n, 3, device=pts.device, dtype=_f32)
        if want_vol and n >= kernels.k2_bricks_min:
            scratch = _bricks_scratch(n, pts.device)
            L.call("gens_lookup_volume_bwd2_bricks", vs.table, vs.dim_table, vs.n, layout, L.ptr(pts), L.ptr(g_out), L.ptr(_c(gg_pts.detach())),
                   L.ptr_table(ggv), n, L.ptr(gg_out), L.ptr_table(g_vols2), L.ptr(g_pts2), L.ptr(scratch, torch.uint8), scratch.numel(), L.stream(),
                   nbytes=n * (36 + 32 * vs.n), label="gens_lookup_volume_bwd2")
        else:
            L.call("gens_lookup_volume_bwd2", vs.table, vs.dim_table, vs.n, layout, L.ptr(pts), L.ptr(g_out), L.ptr(_c(gg_pts.detach())),
                   L.ptr_table(ggv), n, L.ptr(gg_out), L.ptr_table(g_vols2), L.ptr(g_pts2), L.stream(), nbytes=n * (36 + 32 * vs.n))
        # outputs are plain tensors: third order through the sampler is dropped, as in the reference (cuda_gridsample.py:110-123)
        if want_vol:
            gv = tuple(g.reshape(v.shape) for g, v in zip(g_vols2, vols))
        else:
            gv = tuple(None for _ in vols)
        return (gg_out, g_pts2, None, None) + gv


def lookup_volume(pts, volumes):
    """pts (N,3), volumes: list of (1,4,X,Y,Z) tensors (planar) or a packed VolumeSet -> (N, 4L).  Twice differentiable."""
    pts = pts.reshape(-1, 3)
    if isinstance(volumes, VolumeSet):
        return _Lookup.apply(pts, volumes.layout, *volumes.tensors)
    return _Lookup.apply(pts, L.LAYOUT_PLANAR, *[_c(v.to(_f32)) for v in volumes])


# ------------------------------------------------------------------------------------------------------------------
# K3  nearest visibility look-up, ray point generation
# ------------------------------------------------------------------------------------------------------------------
def lookup_mask(pts, masks, return_values=False, out=None):
    """-> valid (N,) bool [, values (N,L) float]: lookup_volume(pts, mask_volumes, 'nearest') (projector.py:231,240).
    out: optional (N,) uint8 buffer (a slice of a step's flag array) the flags are written to."""
    ms = masks if isinstance(masks, VolumeSet) else VolumeSet.masks(masks)
    pts = _c(pts.detach().reshape(-1, 3).to(_f32))
    n = pts.shape[0]
    valid = out if out is not None else torch.empty(n, device=pts.device, dtype=torch.uint8)
    vals = torch.empty(n, ms.n, device=pts.device, dtype=_f32) if return_values else None
    L.call("gens_lookup_mask_nearest", ms.table, ms.dim_table, ms.n, L.ptr(pts), n, L.ptr(valid, torch.uint8), L.ptr(vals), L.stream())
    return (valid.bool(), vals) if return_values else valid.bool()


def compact_valid(valid):
    """Device-side `nonzero` with the reference's rescue (no valid point -> the first 10): -> (idx (N,) int64, count (1,) int32).
    Only idx[:count] is meaningful; nothing is copied to the host, so the caller never synchronises."""
    v = _c(valid.reshape(-1))
    v = v.view(torch.uint8) if v.dtype == torch.bool else v.to(torch.uint8)
    n = v.shape[0]
    idx = torch.empty(max(n, 10), device=v.device, dtype=torch.int64)
    count = torch.empty(1, device=v.device, dtype=torch.int32)
    scratch = torch.empty((n + 1023) // 1024 + 1, device=v.device, dtype=torch.int32)
    L.call("gens_compact_valid", L.ptr(v, torch.uint8), n, L.ptr(idx, torch.int64), L.ptr(count, torch.int32), L.ptr(scratch, torch.int32),
           L.stream(), nbytes=n * 9)
    return idx, count


def coarse_z(near, far, steps, t_rand, b):
    """z (B, n) = near + (far - near) * steps[None, :] (+ (t_rand - 0.5) * 2 / n): implicit_surface.py:356-363 in one launch.
    near / far: (1, 1) for the whole batch or (B, 1) per ray; steps (n,) = linspace(0, 1, n); t_rand (B, 1) on the device or None."""
    n = steps.shape[0]
    nr, fr = _c(near.detach().to(_f32).reshape(-1)), _c(far.detach().to(_f32).reshape(-1))
    assert nr.numel() == fr.numel() and nr.numel() in (1, b), "near / far: one value or one per ray"
    z = torch.empty(b, n, device=steps.device, dtype=_f32)
    tr = None if t_rand is None else _c(t_rand.detach().to(_f32).reshape(-1))
    L.call("gens_coarse_z", L.ptr(nr), L.ptr(fr), 1 if nr.numel() == b and b > 1 else 0, L.ptr(_c(steps)), L.ptr(tr), b, n, L.ptr(z), L.stream())
    return z


def compact_fill(valid, sdf=None, grad=None, rgb=None, vis=None):
    """compact_valid in TWO launches (gens_compact_points) that also write the reference's values for the unselected rows into the given
    dense outputs (Q8: sdf 100, gradient / colour 0, no visible source view) -- no torch.full / zeros before the network launches.
    -> (idx (N,) int64, count (1,) int32)."""
    v = _c(valid.reshape(-1))
    v = v.view(torch.uint8) if v.dtype == torch.bool else v.to(torch.uint8)
    n = v.shape[0]
    dev = v.device
    idx = torch.empty(max(n, 10), device=dev, dtype=torch.int64)
    counts = torch.empty(3, device=dev, dtype=torch.int32)
    scratch = torch.empty(L.load().gens_compact_points_scratch(n), device=dev, dtype=torch.int32)
    L.call("gens_compact_points", L.ptr(v, torch.uint8), n, 0, n, L.ptr(idx, torch.int64), L.ptr(counts, torch.int32), L.ptr(sdf), L.ptr(grad), None,
           L.ptr(rgb), L.ptr(vis, torch.uint8), 0 if vis is None else vis.shape[-1], None, 0, None, None, L.ptr(scratch, torch.int32), L.stream(),
           nbytes=n * 9, label="gens_compact_valid")
    return idx, counts[0:1]


def _mask_args(masks):
    """(pointer table, dims, levels, mask_bits): a scene's VolumeSet is read through its bit-packed copy (built once)."""
    if isinstance(masks, VolumeSet):
        return masks.bit_table(), masks.dim_table, masks.n, 1
    ms = VolumeSet.masks(masks)
    return ms.table, ms.dim_table, ms.n, 0


def ray_points(rays_o, rays_d, z, masks, mid=False, sample_dist=0.0, out=None):
    """pts (B*n,3) = o + d * (z or section mid-points), valid (B*n,) bool.  out: optional (pts, valid uint8) buffers to write (slices of a
    step's point / flag arrays)."""
    table, dims, nl, bits = _mask_args(masks)
    b, n = z.shape
    pts, valid = out if out is not None else (torch.empty(b * n, 3, device=z.device, dtype=_f32), torch.empty(b * n, device=z.device, dtype=torch.uint8))
    L.call("gens_ray_points", L.ptr(_c(rays_o)), L.ptr(_c(rays_d)), L.ptr(_c(z)), b, n, 1 if mid else 0, float(sample_dist), table,
           dims, nl, bits, L.ptr(pts), L.ptr(valid, torch.uint8), L.stream(), nbytes=b * n * 17 + b * 24)
    return pts, valid.view(torch.bool)


# ------------------------------------------------------------------------------------------------------------------
# K4  lookup_feature (projector.py:294-349)
# ------------------------------------------------------------------------------------------------------------------
class _LookupFeature(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, w2c, intr, c2w, imgs_tex, *feat_tex):
        nv = imgs_tex.shape[0]
        nl = len(feat_tex)
        n = pts.shape[0]
        s = nv - 1
        hw = [d for f in feat_tex for d in f.shape[1:3]]
        assert tuple(imgs_tex.shape[1:3]) == tuple(feat_tex[0].shape[1:3]), "RGB images must match feature level 0"
        out = torch.empty(n, s, 3 + 4 * nl, device=pts.device, dtype=_f32)
        ray_diff = torch.empty(n, s, 4, device=pts.device, dtype=_f32)
        vis = torch.empty(n, s, device=pts.device, dtype=torch.uint8)
        feats = [aligned16(f.detach()) for f in feat_tex]
        L.call("gens_lookup_feature_fwd", L.ptr_table(feats, align=16), L.int_table(hw), nl, L.ptr(aligned16(imgs_tex.detach()), align=16), L.ptr(w2c), L.ptr(intr),
               L.ptr(c2w), nv, L.ptr(pts), n, L.ptr(out), L.ptr(ray_diff), L.ptr(vis, torch.uint8), L.stream(),
               nbytes=n * 12 + n * s * (4 * (3 + 4 * nl) + 17))
        ctx.save_for_backward(pts, w2c, intr)
        ctx.meta = (nv, hw, [f.shape for f in feat_tex], imgs_tex.shape)
        ctx.mark_non_differentiable(ray_diff, vis)
        return out, ray_diff, vis

    @staticmethod
    def backward(ctx, g_out, _g_rd, _g_vis):
        pts, w2c, intr = ctx.saved_tensors
        nv, hw, fshapes, ishape = ctx.meta
        nl = len(fshapes)
        want_img = ctx.needs_input_grad[4]
        want_feat = any(ctx.needs_input_grad[5:])
        g_feats = [torch.zeros(s, device=pts.device, dtype=_f32) for s in fshapes] if want_feat else None
        g_imgs = torch.zeros(ishape, device=pts.device, dtype=_f32) if want_img else None
        if want_feat or want_img:
            L.call("gens_lookup_feature_bwd", L.int_table(hw), nl, L.ptr(w2c), L.ptr(intr), nv, L.ptr(pts), L.ptr(_c(g_out)), pts.shape[0],
                   L.ptr_table(g_feats), L.ptr(g_imgs), L.stream())
        return (None, None, None, None, g_imgs) + (tuple(g_feats) if want_feat else tuple(None for _ in fshapes))


class SceneViews:
    """Per-scene camera matrices + texel copies of the images and the feature pyramid (built once per scene)."""

    def __init__(self, imgs, intrs, c2ws, features):
        self.nv = imgs.shape[0]
        self.cams = SceneCams.of(intrs, c2ws)
        self.c2w, self.w2c, self.intr = self.cams.c2w, self.cams.w2c, self.cams.intr
        self.imgs_tex, *self.feat_tex = pack_maps([imgs, *features])


def lookup_feature(pts, views):
    """-> feat_views (N,S,3+4L), ray_diff (N,S,4), mask (N,S) bool.  Differentiable w.r.t. images / features."""
    pts = _c(pts.detach().reshape(-1, 3).to(_f32))
    out, rd, vis = _LookupFeature.apply(pts, views.w2c, views.intr, views.c2w, views.imgs_tex, *views.feat_tex)
    return out, rd, vis.bool()


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]      # private helpers travel too: the package namespace is the old module's
