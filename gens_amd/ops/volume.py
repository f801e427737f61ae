"""K1: Volume.agg_mean_var (volume.py:13-63), forward and backward, one level or a scene's pyramid.

Part of gens_amd.ops (see ops/__init__.py); citations are relative to /root/reference."""
from .base import *  # noqa: F401,F403

# ------------------------------------------------------------------------------------------------------------------
# K1  Volume.agg_mean_var (volume.py:13-63)
# ------------------------------------------------------------------------------------------------------------------
class _VolumeBuild(torch.autograd.Function):
    """One level through gens_volume_build_fwd / gens_volume_build_bwd (the wave-window backward): the measurement scripts' and the cross-checks'
    single-level form; a scene's pyramid goes through _VolumeBuildLevels."""

    @staticmethod
    def forward(ctx, feat_tex, w2c, intr, scale, d, min_vis):
        nv, h, w, cp = feat_tex.shape
        assert cp == 4, "volume build expects 4-channel feature levels (confs/gens.conf:60-62)"
        vol = torch.empty(1, 8, d, d, d, device=feat_tex.device, dtype=_f32)
        mask = torch.empty(1, 1, d, d, d, device=feat_tex.device, dtype=_f32)
        L.call("gens_volume_build_fwd", L.ptr(aligned16(feat_tex), align=16), L.ptr(w2c), L.ptr(intr), scale, nv, h, w, d, min_vis, L.ptr(vol),
               L.ptr(mask), L.stream(), nbytes=nv * h * w * 16 + 36 * d ** 3)
        ctx.save_for_backward(feat_tex, w2c, intr)
        ctx.meta = (scale, d)
        ctx.mark_non_differentiable(mask)
        return vol, mask

    @staticmethod
    def backward(ctx, g_vol, _g_mask):
        feat_tex, w2c, intr = ctx.saved_tensors
        scale, d = ctx.meta
        return _volume_build_bwd(feat_tex, w2c, intr, scale, d, g_vol), None, None, None, None, None


def _volume_build_bwd(feat_tex, w2c, intr, scale, d, g_vol):
    """d(volume)/d(texels) of one level with the wave-window kernel (gens_volume_build_bwd): what the all-level image-tile kernel does not cover
    (volume sides that are not multiples of 16), and its cross-check (kernels.k1_bwd = "window")."""
    nv, h, w, _ = feat_tex.shape
    g = torch.zeros_like(feat_tex)
    L.call("gens_volume_build_bwd", L.ptr(_c(feat_tex)), L.ptr(w2c), L.ptr(intr), scale, nv, h, w, d, L.ptr(_c(g_vol)), L.ptr(g),
           L.stream(), nbytes=2 * nv * h * w * 16 + 32 * d ** 3)                  # texels read + their gradient written, 8 cotangent planes read
    return g


class _VolumeBuildLevels(torch.autograd.Function):
    """All levels of a scene in one launch (gens_volume_build_levels); the backward of all levels in one launch set
    (gens_volume_build_bwd_levels), from the means and visible-view counts the forward pass leaves."""

    @staticmethod
    def forward(ctx, w2c, dims, min_vis, *tex_and_intr):
        n = len(dims)
        texs, intrs = tex_and_intr[:n], tex_and_intr[n:]
        dev = w2c.device
        nv = texs[0].shape[0]
        vols = [torch.empty(1, 8, d, d, d, device=dev, dtype=_f32) for d in dims]
        masks = [torch.empty(1, 1, d, d, d, device=dev, dtype=_f32) for d in dims]
        hw = [x for t in texs for x in (t.shape[1], t.shape[2])]
        for t in texs:
            assert t.shape[0] == nv and t.shape[3] == 4, "volume build expects 4-channel feature levels (confs/gens.conf:60-62)"
        texs_c = [aligned16(t) for t in texs]
        want = any(ctx.needs_input_grad[3:3 + n])
        levels_bwd = want and kernels.k1_bwd == "auto" and L.load().gens_volume_build_bwd_levels_scratch_bytes(L.int_table(hw), L.int_table(dims), n, nv) > 0
        counts = [torch.empty(d ** 3, device=dev, dtype=torch.uint8) for d in dims] if levels_bwd else None
        # the masks as bits too (what the ray-point / nearest look-up kernels read): written by the same launch, not packed from the float planes
        # by a launch per level and step (VolumeSet.bit_table finds them on the mask tensors)
        bits = [torch.empty((d ** 3 + 31) // 32, device=dev, dtype=torch.int32) for d in dims]
        L.call("gens_volume_build_levels_bits", L.ptr_table(texs_c), L.int_table(hw), L.int_table(dims), n, L.ptr(w2c), L.ptr_table(list(intrs)), nv, min_vis,
               L.ptr_table(vols), L.ptr_table(masks), L.ptr_table(counts, torch.uint8), L.ptr_table(bits, torch.int32), L.stream(),
               nbytes=sum(nv * t.shape[1] * t.shape[2] * 16 + 36 * d ** 3 for t, d in zip(texs, dims)), label="gens_volume_build_levels")
        ctx.save_for_backward(w2c, *texs, *intrs, *(vols + counts if levels_bwd else []))
        ctx.dims = list(dims)
        ctx.levels_bwd = levels_bwd
        ctx.mark_non_differentiable(*masks, *bits)
        return (*vols, *masks, *bits)

    @staticmethod
    def backward(ctx, *grads):
        n = len(ctx.dims)
        w2c, rest = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        texs, intrs = rest[:n], rest[n:2 * n]
        on = [grads[l] is not None and ctx.needs_input_grad[3 + l] for l in range(n)]
        if ctx.levels_bwd and any(on):
            vols, counts = rest[2 * n:3 * n], rest[3 * n:4 * n]
            nv = texs[0].shape[0]
            hw = [x for t in texs for x in (t.shape[1], t.shape[2])]
            texs_c = [aligned16(t) for t in texs]
            g_vols = [aligned16(_c(grads[l])) if on[l] else None for l in range(n)]
            # one zeroed buffer for all levels' gradients (one fill), handed out as per-level views
            sizes = [t.numel() if on[l] else 0 for l, t in enumerate(texs)]
            flat = torch.zeros(sum(sizes), device=w2c.device, dtype=_f32)
            out, at = [], 0
            for l, t in enumerate(texs):
                out.append(flat[at:at + sizes[l]].view(t.shape) if on[l] else None)
                at += sizes[l]
            need = L.load().gens_volume_build_bwd_levels_scratch_bytes(L.int_table(hw), L.int_table(ctx.dims), n, nv)
            scratch = torch.empty(need, device=w2c.device, dtype=torch.uint8)
            L.call("gens_volume_build_bwd_levels", L.ptr_table(texs_c), L.int_table(hw), L.int_table(ctx.dims), n, L.ptr(w2c), L.ptr_table(list(intrs)), nv,
                   L.ptr_table(list(vols)), L.ptr_table(list(counts), torch.uint8), L.ptr_table(g_vols), L.ptr_table(out), L.ptr(scratch, torch.uint8), need,
                   L.stream(), nbytes=sum(2 * nv * t.shape[1] * t.shape[2] * 16 + 32 * d ** 3 for t, d, o in zip(texs, ctx.dims, on) if o),      # texels read + their gradient written, 8 cotangent planes read (the means and counts it also reads are the design's, not the algorithm's)
                   label="gens_volume_build_bwd")
            return (None, None, None, *out, *([None] * n))
        out = []
        for l, d in enumerate(ctx.dims):
            out.append(_volume_build_bwd(texs[l], w2c, intrs[l], 1.0, d, grads[l]) if on[l] else None)
        return (None, None, None, *out, *([None] * n))


def volume_build(features, intrs, c2ws, dims, min_vis_view=1):
    """features: list of (nv,4,H_i,W_i) NCHW -> (volumes [(1,8,D,D,D)], masks [(1,1,D,D,D)]).  One launch for all levels."""
    # inverse(c2ws) and the intrinsics with rows 0-1 times 0.5^lvl (Q2; an exact power-of-two scaling, the product the reference forms per
    # level, volume.py:24-25) come from the scene's one set-up launch; the texel copies of the maps from one layout launch
    cams = SceneCams.of(intrs, c2ws)
    texs = pack_maps([features[lvl] for lvl in range(len(dims))])
    out = _VolumeBuildLevels.apply(cams.w2c, [int(d) for d in dims], int(min_vis_view), *texs, *cams.ks[:len(dims)])
    n = len(dims)
    masks = list(out[n:2 * n])
    for m, words in zip(masks, out[2 * n:]):
        try:
            m._gens_bits = (m._version, words)                      # (VolumeSet.bit_table)
        except (AttributeError, RuntimeError):
            pass
    return list(out[:n]), masks


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]      # private helpers travel too: the package namespace is the old module's
