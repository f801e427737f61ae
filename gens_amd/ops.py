"""Autograd-aware operators of the GenS hot path, each a thin shim over one C-ABI entry point of libgens_hip.so.

The shims allocate outputs with torch (device memory + current stream only) and wire first / second order
derivatives the way the reference's Function pair does (models/modules/grid_sample_cuda/cuda_gridsample.py:71-123):
twice differentiable, outputs of the second backward are constants.  Citations are relative to /root/reference.
"""
import ctypes as C
import os
import math

import torch

from . import lib as L

_f32 = torch.float32


class KernelChoice:
    """Which generation of a fused kernel an operator launches -- plain attributes, set once from the environment at import (the switches
    of INTEGRATION.md) and changed by assignment afterwards (tests: monkeypatch.setattr(ops.kernels, ...)).  The operators read these
    attributes; nothing on a launch path reads os.environ.
        sdf_value / sdf_grad   "transposed" (k6t / k6g: register-chained, the default) | "rowmajor" (k6_sdfmlp.hip: cross-check, other shapes)
        blend                  "transposed" (k7t, two to four source views) | "rowmajor" (k7_blend.hip)
        blend_train_fwd        "transposed" (the training step's forward through k7t + gens_blend_pack_t) | "rowmajor" (k18's own forward)
        k1_bwd                 "auto" (all levels on the image-tile kernel) | "window" (the wave-window kernel, level by level)
        tex_cache              texel copies kept on the map tensors (pack_maps)"""

    def __init__(self, env=os.environ):
        self.sdf_value = "rowmajor" if env.get("GENS_SDF_VALUE_ROWMAJOR") else "transposed"
        self.sdf_grad = "rowmajor" if env.get("GENS_SDF_GRAD_ROWMAJOR") else "transposed"
        self.blend = "rowmajor" if env.get("GENS_BLEND_ROWMAJOR") else "transposed"
        self.blend_train_fwd = "rowmajor" if env.get("GENS_BLEND_TRAIN_ROWMAJOR") else "transposed"
        self.k1_bwd = "window" if env.get("GENS_K1_BWD_WINDOW") else "auto"
        self.tex_cache = not env.get("GENS_NO_TEX_CACHE")


kernels = KernelChoice()


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


_SDF_GRAD_STASH = {}


def sdf_grad_stash(device):
    """gens_sdf_grad's SIMD-private slots (softplus' of one layer between the forward and the reverse chain): one zeroed buffer per device for the
    life of the process, shared by every call."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    buf = _SDF_GRAD_STASH.get(key)
    if buf is None:
        buf = _SDF_GRAD_STASH[key] = torch.zeros(L.load().gens_sdf_grad_stash_bytes(), device=torch.device("cuda", key), dtype=torch.uint8)
    return buf


def aligned16(t):
    """Contiguous and 16-byte aligned (what the float4 / float2 accesses of the texel, packed-volume, K15 and K16 kernels need): a
    contiguous VIEW that starts mid-allocation (flat[1:].view(c, n)) is copied; everything torch allocates itself already qualifies."""
    t = _c(t)
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _dev_f32(t, device):
    return _c(t.to(device=device, dtype=_f32))


def inv(a):
    """torch.linalg.inv without its error check: the same LU solve, but the `info` read-back of linalg.inv is a device-to-host copy
    that drains the stream (three of them per training step: camera poses, intrinsics, the reference rotation)."""
    return torch.linalg.inv_ex(a).inverse


# ------------------------------------------------------------------------------------------------------------------
# texel layout (NHWC, channels padded to a multiple of 4)
# ------------------------------------------------------------------------------------------------------------------
class _PackNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        ctx.shape = (n, c, h, w)
        out = torch.empty(n, h, w, 4 * ((c + 3) // 4), device=x.device, dtype=_f32)
        L.call("gens_pack_nchw", L.ptr(_c(x)), L.ptr(out), n, c, h, w, L.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        out = torch.empty(n, c, h, w, device=g.device, dtype=_f32)
        L.call("gens_unpack_nhwc", L.ptr(_c(g)), L.ptr(out), n, c, h, w, L.stream())
        return out


def pack_nchw(x):
    """(n,C,H,W) -> (n,H,W,C_pad) texels; differentiable."""
    return _PackNCHW.apply(x)


class _PackMaps(torch.autograd.Function):
    """gens_pack_nchw for several maps in ONE launch (gens_pack_maps); the backward unpacks the gradients that arrived in one launch too."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [_c(x) for x in xs]
        outs = [torch.empty(x.shape[0], x.shape[2], x.shape[3], 4 * ((x.shape[1] + 3) // 4), device=x.device, dtype=_f32) for x in xs]
        nchw = [d for x in xs for d in x.shape]
        L.call("gens_pack_maps", L.ptr_table(xs), L.ptr_table(outs, align=16), L.int_table(nchw), len(xs), L.stream())
        ctx.shapes = [tuple(x.shape) for x in xs]
        ctx.set_materialize_grads(False)          # a map nothing downstream differentiates gets no gradient pass (not a pass over zeros)
        # the texels of a map that needs no gradient must not hang on this node: they are kept on the map (pack_maps) and outlive the step,
        # while the node's other inputs (this step's feature maps) do not
        ctx.mark_non_differentiable(*[o for k, o in enumerate(outs) if not ctx.needs_input_grad[k]])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        live = [k for k, g in enumerate(gs) if g is not None and ctx.needs_input_grad[k]]
        res = [None] * len(gs)
        if live:
            srcs = [aligned16(gs[k]) for k in live]
            dsts = [torch.empty(ctx.shapes[k], device=srcs[0].device, dtype=_f32) for k in live]
            nchw = [d for k in live for d in ctx.shapes[k]]
            L.call("gens_unpack_maps", L.ptr_table(srcs, align=16), L.ptr_table(dsts), L.int_table(nchw), len(live), L.stream())
            for k, d in zip(live, dsts):
                res[k] = d
        return tuple(res)


def pack_maps(maps):
    """[(n,C,H,W), ...] -> their (n,H,W,C_pad) texel copies, in one launch for the maps that have none yet.  The texels of a map are kept ON
    the map tensor (attribute `_gens_tex`, valid for the tensor's current version and autograd mode): the volume build and the renderer of one
    forward pass (volume.py:21-61 and projector.py:294-349 read the same `features`) share one layout pass, and a frozen map (fine-tuning:
    GenS.features) is packed once, not once per step."""
    maps = [m if m.dtype == _f32 else m.to(_f32) for m in maps]
    out, todo = [None] * len(maps), []
    grad_mode = torch.is_grad_enabled()
    for k, m in enumerate(maps):
        hit = getattr(m, "_gens_tex", None) if kernels.tex_cache else None
        if hit is not None and hit[0] == m._version and hit[1] == (grad_mode and m.requires_grad) and hit[2].device == m.device:
            out[k] = hit[2]
        else:
            todo.append(k)
    for s in range(0, len(todo), 8):
        part = todo[s:s + 8]
        texs = _PackMaps.apply(*[maps[k] for k in part])
        for k, t in zip(part, texs):
            out[k] = t
            try:
                maps[k]._gens_tex = (maps[k]._version, grad_mode and maps[k].requires_grad, t)
            except (AttributeError, RuntimeError):
                pass
    return out


class SceneCams:
    """The camera constants of a scene on the device, from ONE launch (gens_scene_setup): w2c = inverse(c2ws), the per-level intrinsics,
    inverse(c2ws[0,:3,:3]) and inverse(intrs)[0,:3,:3] (volume.py:24-25,34; projector.py:317-322,364; implicit_surface.py:242,245).  The
    reference calls torch.inverse at each of those places: a batched LU of ~11 launches each, 45 launches per training step.
    `SceneCams.of` returns the instance built for the same two tensor OBJECTS at their current versions (the volume build and the renderer
    of one forward pass receive the same `intrs` / `c2ws`)."""
    _last = None

    def __init__(self, intrs, c2ws):
        dev = c2ws.device
        self.nv = int(c2ws.shape[0])
        self.c2w, self.intr = _dev_f32(c2ws.detach(), dev), _dev_f32(intrs.detach(), dev)
        self.buf = torch.empty(L.load().gens_scene_cams_floats(self.nv), device=dev, dtype=_f32)
        L.call("gens_scene_setup", L.ptr(self.c2w), L.ptr(self.intr), self.nv, L.ptr(self.buf), L.stream())
        nv, o = self.nv, self.nv * 16
        self.w2c = self.buf[:o].view(nv, 4, 4)
        self.ks = [self.buf[o + l * nv * 16:o + (l + 1) * nv * 16].view(nv, 4, 4) for l in range(L.MAX_LEVELS)]
        o += L.MAX_LEVELS * nv * 16
        self.rot_inv = self.buf[o:o + 9]                     # row-major inverse(c2ws[0, :3, :3])
        self.kinv_ref = self.buf[o + 12:o + 21].view(3, 3)
        self.status = self.buf[o + 24:o + 25].view(torch.int32)

    @staticmethod
    def of(intrs, c2ws):
        key = (intrs, c2ws, intrs._version, c2ws._version)
        last = SceneCams._last
        if last is not None and last[0][0] is intrs and last[0][1] is c2ws and last[0][2:] == key[2:]:
            return last[1]
        cams = SceneCams(intrs, c2ws)
        SceneCams._last = (key, cams)
        return cams

    def check(self):
        """Raise like torch.inverse does for a singular pose / intrinsics matrix.  Reads one int back: call it where the host synchronises
        anyway (end of validate(), end of a training forward), never between launches."""
        if int(self.status.item()) != 0:
            raise RuntimeError("linalg.inv: a camera pose or intrinsics matrix of the scene is singular (gens_scene_setup)")


def pack_volume(v):
    """(1,4,X,Y,Z) or (4,X,Y,Z) -> (X,Y,Z,4) texels (inference fast path; not differentiable)."""
    v = v.detach()
    if v.dim() == 5:
        v = v[0]
    assert v.shape[0] == 4, "only 4-channel volume levels are supported"
    _, x, y, z = v.shape
    out = torch.empty(x, y, z, 4, device=v.device, dtype=_f32)
    L.call("gens_pack_volume", L.ptr(_c(v)), L.ptr(out), x, y, z, L.stream())
    return out


class VolumeSet:
    """A pyramid of volumes as the kernels see it: host pointer table + dims + layout."""

    def __init__(self, tensors, layout):
        self.layout = layout
        self.tensors = [aligned16(t) if layout == L.LAYOUT_PACKED else _c(t) for t in tensors]
        if layout == L.LAYOUT_PACKED:
            dims = [tuple(t.shape[:3]) for t in self.tensors]
        else:
            for t in self.tensors:
                assert t.shape[-4] == 4, "only 4-channel volume levels are supported (confs/gens.conf:63-67)"
            dims = [tuple(t.shape[-3:]) for t in self.tensors]
        self.dims = dims
        self.n = len(self.tensors)
        assert 1 <= self.n <= L.MAX_LEVELS
        self.table = L.ptr_table(self.tensors)
        self.dim_table = L.int_table([d for dd in dims for d in dd])

    @staticmethod
    def packed(volumes):
        return VolumeSet([pack_volume(v) for v in volumes], L.LAYOUT_PACKED)

    @staticmethod
    def masks(mask_volumes):
        """Mask pyramid (1,1,X,Y,Z) floats for the nearest look-up (K3)."""
        vs = VolumeSet.__new__(VolumeSet)
        vs.layout = L.LAYOUT_PLANAR
        vs.tensors = [_c(m.detach().reshape(m.shape[-3:])) for m in mask_volumes]
        vs.sources = list(mask_volumes)
        vs.dims = [tuple(t.shape) for t in vs.tensors]
        vs.n = len(vs.tensors)
        vs.table = L.ptr_table(vs.tensors)
        vs.dim_table = L.int_table([d for dd in vs.dims for d in dd])
        vs._bits = None
        return vs

    def bit_table(self):
        """Bit-packed copy of a mask pyramid (built once, on first use): HOST pointer table for mask_bits=1 calls.  The words are kept ON the
        mask tensor (valid for its current version): a frozen pyramid (fine-tuning: GenS.mask_volmes) is packed once, not once per step."""
        if getattr(self, "_bits", None) is None:
            words = []
            for t, src in zip(self.tensors, getattr(self, "sources", self.tensors)):
                hit = getattr(src, "_gens_bits", None)
                if hit is not None and hit[0] == src._version and hit[1].device == t.device:
                    words.append(hit[1])
                    continue
                n = t.numel()
                w = torch.empty((n + 31) // 32, device=t.device, dtype=torch.int32)
                L.call("gens_pack_mask_bits", L.ptr(t), n, L.ptr(w, torch.int32), L.stream())
                words.append(w)
                try:
                    src._gens_bits = (src._version, w)
                except (AttributeError, RuntimeError):
                    pass
            self._bits = (words, L.ptr_table(words, torch.int32))
        return self._bits[1]


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
        L.call("gens_volume_build_levels", L.ptr_table(texs_c), L.int_table(hw), L.int_table(dims), n, L.ptr(w2c), L.ptr_table(list(intrs)), nv, min_vis,
               L.ptr_table(vols), L.ptr_table(masks), L.ptr_table(counts, torch.uint8), L.stream(),
               nbytes=sum(nv * t.shape[1] * t.shape[2] * 16 + 36 * d ** 3 for t, d in zip(texs, dims)))
        ctx.save_for_backward(w2c, *texs, *intrs, *(vols + counts if levels_bwd else []))
        ctx.dims = list(dims)
        ctx.levels_bwd = levels_bwd
        ctx.mark_non_differentiable(*masks)
        return (*vols, *masks)

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
                   L.stream(), nbytes=sum(2 * nv * t.shape[1] * t.shape[2] * 16 + 49 * d ** 3 for t, d, o in zip(texs, ctx.dims, on) if o),
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
    return list(out[:n]), list(out[n:])


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


class _LookupBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g_out, pts, layout, want_vol, *vols):
        vs = _vset(layout, [v.detach() for v in vols])
        n = pts.shape[0]
        g_out_c = _c(g_out.detach().to(_f32))
        pts_c = _c(pts.detach().to(_f32))
        g_pts = torch.empty(n, 3, device=pts.device, dtype=_f32)
        g_vols = [torch.zeros_like(v) for v in vs.tensors] if want_vol else None
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


# ------------------------------------------------------------------------------------------------------------------
# K8  compositing (implicit_surface.py:160-168, 202-303)
# ------------------------------------------------------------------------------------------------------------------
def _composite_in(rays_o, rays_d, z, sdf, grad, color, smooth, voxel_mask, src_vis, inv_s, z_max, sample_dist, cos_anneal, rot):
    ci = L.CompositeIn()
    ci.rays_o, ci.rays_d, ci.z = L.ptr(rays_o), L.ptr(rays_d), L.ptr(z)
    ci.sdf, ci.grad, ci.color, ci.smooth = L.ptr(sdf), L.ptr(grad), L.ptr(color), L.ptr(smooth)
    ci.voxel_mask = L.ptr(voxel_mask, torch.uint8)
    ci.src_vis = L.ptr(src_vis, torch.uint8)
    ci.inv_s, ci.z_max = L.ptr(inv_s), L.ptr(z_max)
    ci.n_rays, ci.n = z.shape
    ci.n_src = src_vis.shape[-1] if src_vis is not None else 0
    ci.sample_dist, ci.cos_anneal = float(sample_dist), float(cos_anneal)
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
                                 float(sample_dist), float(cos_anneal), rot)
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
                            _c(rays_d.to(_f32)), z, vm, sv, z_max, float(sample_dist), float(cos_anneal), rot)
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
    dst = torch.zeros(nv, h, w, cpad, device=f0.device, dtype=_f32)
    off = 0
    for f in levels:
        f = _c(f.detach().to(_f32))
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


# ------------------------------------------------------------------------------------------------------------------
# K11  lattice (implicit_surface.py:407-418)
# ------------------------------------------------------------------------------------------------------------------
def lattice_points(bound_min, bound_max, resolution, first, count, device):
    lo = (C.c_float * 3)(*[float(v) for v in bound_min])
    hi = (C.c_float * 3)(*[float(v) for v in bound_max])
    pts = torch.empty(count, 3, device=device, dtype=_f32)
    L.call("gens_lattice_points", lo, hi, int(resolution), int(first), int(count), L.ptr(pts), L.stream())
    return pts


# ------------------------------------------------------------------------------------------------------------------
# K12  iso-surface extraction (mcubes.marching_cubes at implicit_surface.py:423)
# ------------------------------------------------------------------------------------------------------------------
_MC_TABLES = {}


def marching_cubes(u, threshold=0.0):
    """u (X,Y,Z) float32 device tensor -> (vertices (V,3) float64 in index coordinates, triangles (T,3) int32), both on the
    device.  Classic marching cubes with the case table of gens_amd/mc_tables.py; order as in oracle/mc_oracle.py."""
    from . import mc_tables
    dev = u.device
    if dev not in _MC_TABLES:
        _MC_TABLES[dev] = (torch.from_numpy(mc_tables.TRI_TABLE.copy()).to(dev), torch.from_numpy(mc_tables.TRI_COUNT.copy()).to(dev))
    table, count = _MC_TABLES[dev]
    u = _c(u.detach().to(_f32))
    x, y, z = u.shape
    total = x * y * z
    vmask, vcount, cases, tcount = (torch.empty(total, device=dev, dtype=torch.uint8) for _ in range(4))
    u8 = torch.uint8
    L.call("gens_mc_classify", L.ptr(u), x, y, z, float(threshold), L.ptr(count, u8), L.ptr(vmask, u8), L.ptr(vcount, u8), L.ptr(cases, u8),
           L.ptr(tcount, u8), L.stream(), nbytes=total * 8)
    vend = torch.cumsum(vcount, 0, dtype=torch.int32)
    tend = torch.cumsum(tcount, 0, dtype=torch.int32)
    nv, nt = int(vend[-1]), int(tend[-1])
    vertices = torch.empty(nv, 3, device=dev, dtype=torch.float64)
    triangles = torch.empty(nt, 3, device=dev, dtype=torch.int32)
    if nv == 0:
        return vertices, triangles
    voff = vend - vcount        # exclusive scans
    toff = tend - tcount
    del vend, tend
    i32 = torch.int32
    L.call("gens_mc_emit", L.ptr(u), x, y, z, float(threshold), L.ptr(table, torch.int8), table.shape[1], L.ptr(vmask, u8), L.ptr(voff, i32),
           L.ptr(cases, u8), L.ptr(tcount, u8), L.ptr(toff, i32), L.ptr(vertices, torch.float64),
           L.ptr(triangles, i32) if nt else L.ptr(torch.empty(1, 3, device=dev, dtype=i32), i32), L.stream(),
           nbytes=total * 15 + nv * 24 + nt * 12)
    return vertices, triangles


# ------------------------------------------------------------------------------------------------------------------
# K6  fused SDF network (inference): look-up + encodings + 7 layers on fp32 MFMA (+ d sdf/dx)   (sdf_network.py:98-146)
# ------------------------------------------------------------------------------------------------------------------
def _pack_b_fragments(w):
    """(J, K) matrix -> MFMA 32x32x2 B fragments [ceil(J/32)][ceil(K/2)][64]: lane l of fragment (nt, kk) holds
    w[32 nt + (l & 31)][2 kk + (l >> 5)] (zero padded), so one B operand is one contiguous 256-B load."""
    j, k = w.shape
    nt, kk = (j + 31) // 32, (k + 1) // 2
    wp = torch.zeros(nt * 32, kk * 2, device=w.device, dtype=_f32)
    wp[:j, :k] = w
    return wp.view(nt, 32, kk, 2).permute(0, 2, 3, 1).contiguous()


def _pack_b_groups(w):
    """(J, K) matrix -> grouped fp32 MFMA B stream for gens_sdf_mlp: [ceil(J/32)][ceil(K/8)][64][4]; lane l of group
    (nt, g) holds w[32 nt + (l & 31)][8 g + 4 (l >> 5) + 0..3] (zero padded): one global_load_dwordx4 feeds 4 MFMAs."""
    j, k = w.shape
    nt, g = (j + 31) // 32, (k + 7) // 8
    wp = torch.zeros(nt * 32, g * 8, device=w.device, dtype=_f32)
    wp[:j, :k] = w
    return wp.view(nt, 32, g, 2, 4).permute(0, 2, 3, 1, 4).contiguous()


def _pack_b16(w, groups):
    """(J <= 16, K) matrix -> B stream of a narrow layer for two 16x16x4 fp32 MFMA tiles (k7_blend.hip::narrow_group): for every group
    (k0, S) of 4 S reduction columns, 64 lanes x S floats; lane l holds w[l % 16][k0 + S (l // 16) + 0..S-1] (zero padded)."""
    j, k = w.shape
    assert j <= 16
    kmax = max(k0 + 4 * s for k0, s in groups)
    wp = torch.zeros(16, kmax, device=w.device, dtype=_f32)
    wp[:j, :k] = w
    parts = []
    for k0, s in groups:
        blk = wp[:, k0:k0 + 4 * s].reshape(16, 4, s)           # [j][q][s]
        parts.append(blk.permute(1, 0, 2).reshape(-1))         # lane = q * 16 + j
    return torch.cat(parts).contiguous()


def _value_slots(n_levels):
    """Which input column every B-operand slot of k6v_sdf_value_f16.hip carries: three tables of shape (blocks, half, 8) holding a column
    number, -1 for the constant-one slot and -2 for a zero slot.  Hidden blocks: the accumulator layout of the previous layer (lane half h,
    register r of tile t = feature 32 t + 8 (r >> 2) + 4 h + (r & 3) = slot r & 7 of block 2 t + (r >> 3)).  Point encoding: half 0 holds
    pe[0:15] and the one, half 1 pe[15:27].  Volume features: half 0 the channels of the levels below the middle one and its first two, half
    1 the levels above and its last two; five encodings per channel (column e * CF + channel, sdf_network.py:104-107), then the one."""
    cf = 4 * n_levels
    nch, mid = cf // 2, n_levels // 2
    nc = (5 * nch + 1 + 7) // 8
    hid = torch.tensor([[[32 * (b >> 1) + 16 * (b & 1) + 8 * (s >> 2) + 4 * h + (s & 3) for s in range(8)] for h in range(2)] for b in range(8)])
    pe = torch.full((2, 2, 8), -2, dtype=torch.long)
    for q in range(16):
        pe[q >> 3, 0, q & 7] = q if q < 15 else -1
        if q < 12:
            pe[q >> 3, 1, q & 7] = 15 + q
    cond = torch.full((nc, 2, 8), -2, dtype=torch.long)
    for h in range(2):
        nfull = 4 * mid if h == 0 else 4 * (n_levels - 1 - mid)
        for lc in range(nch):
            ch = (lc if h == 0 else 4 * (mid + 1) + lc) if lc < nfull else 4 * mid + 2 * h + (lc - nfull)
            for e in range(5):
                q = 5 * lc + e
                cond[q >> 3, h, q & 7] = e * cf + ch
    cond[(5 * nch) >> 3, 0, (5 * nch) & 7] = -1
    return hid, pe, cond


def _pack_value_units(ws, bs, n_levels):
    """The weight stream and the output row of gens_sdf_value_f16 (layout and scaling: k6v_sdf_value_f16.hip's header).  ws[l] (out_l, in_l)
    and bs[l] are the effective float32 weights of lin0..lin6.  Returns (units (U, 4, 2, 64, 8) float16, w_out (2, 64 + 8 NC) float32,
    largest magnitude handed to half precision)."""
    dev = ws[0].device
    c = 100.0 / math.log(2.0)
    r2 = 1.0 / math.sqrt(2.0)
    hid, pe, cond = (t.to(dev) for t in _value_slots(n_levels))
    fe = 20 * n_levels

    def block_units(aug, table, offset):
        """aug: (128, K + 2) with the bias in column K and zeros in column K + 1; table entries index aug[:, offset + entry]."""
        k = aug.shape[1] - 2
        cols = torch.where(table >= 0, table + offset, torch.where(table == -1, torch.full_like(table, k), torch.full_like(table, k + 1)))
        g = aug[:, cols.reshape(-1)].reshape(4, 32, *table.shape)               # [tile][m][block][half][slot]
        return g.permute(2, 0, 3, 1, 4).reshape(table.shape[0], 4, 64, 8)        # [block][tile][lane = 32 half + m][slot]

    units = []
    zero = torch.zeros(128, 1, device=dev, dtype=_f32)
    for l in range(6):
        w = torch.zeros(128, ws[l].shape[1], device=dev, dtype=_f32)
        w[:ws[l].shape[0]] = ws[l]
        b = torch.zeros(128, 1, device=dev, dtype=_f32)
        b[:bs[l].shape[0], 0] = bs[l]
        if l == 0:
            units.append(block_units(torch.cat([c * w, c * b, zero], 1), pe, 0))
            continue
        h = w[:, :128].clone()
        if l == 3:                                                               # x = cat([h[:101], pe]) / sqrt(2)   (sdf_network.py:111-112)
            skip = torch.cat([c * r2 * w[:, 101:128], zero, zero], 1)            # the one slot of the point encoding carries nothing here
            h = r2 * h
            h[:, 101:] = 0.0
        aug = torch.cat([h, c * w[:, 128:], c * b, zero], 1)
        units.append(block_units(aug, hid, 0))
        if l == 3:
            units.append(block_units(skip, torch.where(pe == -1, torch.full_like(pe, -2), pe), 0))
        units.append(block_units(aug, cond, 128))
    units = torch.cat(units, 0)
    pad = (-units.shape[0]) % 4                                                  # whole chunks of four units
    if pad:
        units = torch.cat([units, torch.zeros(pad, *units.shape[1:], device=dev, dtype=_f32)], 0)
    hi = units.half()
    lo = (units - hi.float()).half()
    stream = torch.stack([hi, lo], 2).contiguous()                               # [unit][tile][hi, lo][lane][slot]
    w_last = ws[6][0]
    nc = cond.shape[0]
    w_out = torch.zeros(2, 64 + 8 * nc, device=dev, dtype=_f32)
    for hh in range(2):
        feat = torch.tensor([32 * t + 8 * (r >> 2) + 4 * hh + (r & 3) for t in range(4) for r in range(16)], device=dev)
        w_out[hh, :64] = w_last[feat] / c
        tb = cond[:, hh].reshape(-1)
        w_out[hh, 64:] = torch.where(tb >= 0, w_last[(128 + tb).clamp(0, 127 + fe)], torch.zeros_like(tb, dtype=_f32))
    return stream, w_out, float(units.abs().max())


def _value_pairs(n_levels):
    """Slot tables of k6t_sdf_value.hip, (groups, half, 4) each (column number, -1 = the constant one, -2 = zero): an MFMA of group g,
    position i multiplies the weights of the two columns [g, 0, i] and [g, 1, i] with what the two lane halves hold.  Hidden groups
    (t, g'): columns 32 t + 8 g' + 4 half + i (the accumulator layout).  Point encoding: half 0 pe[0:15], half 1 pe[15:27] and the one.
    Volume features: as gens_amd.ops._value_slots, 5 encodings per channel of the half, then the one (half 0)."""
    cf = 4 * n_levels
    nch, mid = cf // 2, n_levels // 2
    gc = (5 * nch + 1 + 3) // 4
    hid = torch.tensor([[[32 * t + 8 * g + 4 * h + i for i in range(4)] for h in range(2)] for t in range(4) for g in range(4)])
    pe = torch.full((4, 2, 4), -2, dtype=torch.long)
    for q in range(15):
        pe[q >> 2, 0, q & 3] = q
        pe[q >> 2, 1, q & 3] = 15 + q if q < 12 else (-1 if q == 12 else -2)
    cond = torch.full((gc, 2, 4), -2, dtype=torch.long)
    for h in range(2):
        nfull = 4 * mid if h == 0 else 4 * (n_levels - 1 - mid)
        for lc in range(nch):
            ch = (lc if h == 0 else 4 * (mid + 1) + lc) if lc < nfull else 4 * mid + 2 * h + (lc - nfull)
            for e in range(5):
                q = 5 * lc + e
                cond[q >> 2, h, q & 3] = e * cf + ch
    cond[(5 * nch) >> 2, 0, (5 * nch) & 3] = -1
    return hid, pe, cond


def _pack_value_stream(ws, bs, n_levels):
    """The float32 weight stream and output row of gens_sdf_value (k6t_sdf_value.hip): per group of four feature pairs and output tile
    T one float4 per lane (m, half) = the weights of row 32 T + m for the group's four columns of that half; columns fed by unscaled
    inputs carry 100 / ln 2 (pre-scaled hidden units), layer 3's hidden columns 1 / sqrt(2), its skip columns both; one zero group is
    appended because the kernel requests the next group before it knows there is none.  -> (stream (NG + 1, 4, 64, 4), w_out)."""
    dev = ws[0].device
    c = 100.0 / math.log(2.0)
    r2 = 1.0 / math.sqrt(2.0)
    hid, pe, cond = (t.to(dev) for t in _value_pairs(n_levels))
    fe = 20 * n_levels

    def groups(aug, table, offset):
        k = aug.shape[1] - 2
        cols = torch.where(table >= 0, table + offset, torch.where(table == -1, torch.full_like(table, k), torch.full_like(table, k + 1)))
        g = aug[:, cols.reshape(-1)].reshape(4, 32, *table.shape)               # [tile][m][group][half][i]
        return g.permute(2, 0, 3, 1, 4).reshape(table.shape[0], 4, 64, 4)        # [group][tile][lane = 32 half + m][i]

    out = []
    zero = torch.zeros(128, 1, device=dev, dtype=_f32)
    for l in range(6):
        w = torch.zeros(128, ws[l].shape[1], device=dev, dtype=_f32)
        w[:ws[l].shape[0]] = ws[l]
        b = torch.zeros(128, 1, device=dev, dtype=_f32)
        b[:bs[l].shape[0], 0] = bs[l]
        if l == 0:
            out.append(groups(torch.cat([c * w, c * b, zero], 1), pe, 0))
            continue
        h = w[:, :128].clone()
        if l == 3:                                                               # x = cat([h[:101], pe]) / sqrt(2)   (sdf_network.py:111-112)
            skip = torch.cat([c * r2 * w[:, 101:128], zero, zero], 1)
            h = r2 * h
            h[:, 101:] = 0.0
        aug = torch.cat([h, c * w[:, 128:], c * b, zero], 1)
        out.append(groups(aug, hid if l != 3 else hid[:13], 0))                  # layer 3 reads features 0..103 only
        if l == 3:
            out.append(groups(skip, torch.where(pe == -1, torch.full_like(pe, -2), pe), 0))
        out.append(groups(aug, cond, 128))
    out.append(torch.zeros(1, 4, 64, 4, device=dev, dtype=_f32))
    stream = torch.cat(out, 0).contiguous()
    w_last = ws[6][0]
    w_out = torch.zeros(2, 64 + 4 * cond.shape[0], device=dev, dtype=_f32)
    for hh in range(2):
        feat = torch.tensor([32 * t + 8 * (r >> 2) + 4 * hh + (r & 3) for t in range(4) for r in range(16)], device=dev)
        w_out[hh, :64] = w_last[feat] / c
        tb = cond[:, hh].reshape(-1)
        w_out[hh, 64:] = torch.where(tb >= 0, w_last[(128 + tb).clamp(0, 127 + fe)], torch.zeros_like(tb, dtype=_f32))
    return stream, w_out


def _pack_grad_stream(ws, bs, n_levels):
    """The weight stream and output row of gens_sdf_grad (k6g_sdf_grad.hip): the forward groups of _pack_value_stream, then the reverse
    pass on the TRUE (unscaled) transposed matrices, layer 5 down to 1: 16 groups (layer 2: 13) of W_l[:, :128]^T for the hidden-unit
    gradients, then per pair of conditioning tiles 8 groups (layer 2: 7) of 2 tiles x 8 pairs whose ROWS are ordered so that lane half h,
    register r of tile c receives the gradient of that half's slot 16 c + r, at layer 3 four groups of 1 tile x 16 pairs for the
    point-encoding slots, and after layer 1 the same four groups of W_0^T; two trailing zero groups (the kernel reads two groups ahead)."""
    dev = ws[0].device
    r2 = 1.0 / math.sqrt(2.0)
    c = 100.0 / math.log(2.0)
    hid, pe, cond = (t.to(dev) for t in _value_pairs(n_levels))
    nch = 2 * n_levels
    tc = (5 * nch + 15) // 16
    fwd, _ = _pack_value_stream(ws, bs, n_levels)
    out = [fwd[:-1]]

    def groups(mat, table):
        """mat (32 NT, 128): rows = output rows of NT tiles, columns = hidden units of the layer -> (G, NT, 64, 4)."""
        nt = mat.shape[0] // 32
        g = mat[:, table.reshape(-1)].reshape(nt, 32, *table.shape)
        return g.permute(2, 0, 3, 1, 4).reshape(table.shape[0], nt, 64, 4)

    # accumulator row m of a tile <-> (lane half, register): m = 8 (r >> 2) + 4 half + (r & 3)
    m = torch.arange(32, device=dev)
    row_half, row_reg = (m >> 2) & 1, ((m >> 3) << 2) | (m & 3)
    cond_flat = cond.permute(1, 0, 2).reshape(2, -1)                     # [half][slot] -> feature column, -1 one, -2 nothing
    pe_flat = pe.permute(1, 0, 2).reshape(2, -1)
    for l in range(5, 0, -1):
        w = torch.zeros(128, ws[l].shape[1], device=dev, dtype=_f32)
        w[:ws[l].shape[0]] = ws[l]
        wt = w[:, :128].t().clone()                                       # rows: hidden inputs, columns: units of layer l
        if l == 3:
            wt = r2 * wt
            wt[101:] = 0.0
        out.append(groups(wt, hid if l != 2 else hid[:13]).reshape(-1, 4, 64, 4))
        mc = torch.zeros(32 * tc, 128, device=dev, dtype=_f32)
        for cc in range(tc):
            slot = 16 * cc + row_reg
            col = torch.where(slot < cond_flat.shape[1], cond_flat[row_half, slot.clamp(max=cond_flat.shape[1] - 1)], torch.full_like(slot, -2))
            live = col >= 0
            mc[32 * cc + m[live]] = w[:, 128 + col[live]].t()
        full = groups(mc, hid)                                            # (16 = (t, g), tc, 64, 4)
        for cc in range(0, tc, 2):
            for t in range(4 if l != 2 else 3):
                for gg in range(2):
                    a, b = full[4 * t + 2 * gg], full[4 * t + 2 * gg + 1]
                    out.append(torch.stack([a[cc], b[cc], a[cc + 1], b[cc + 1]])[None])
            if l == 2:
                a, b = full[12], full[13]
                out.append(torch.stack([a[cc], b[cc], a[cc + 1], b[cc + 1]])[None])
        if l == 3 or l == 1:
            src = r2 * w[:, 101:128] if l == 3 else None
            if l == 1:
                w0 = torch.zeros(128, 27, device=dev, dtype=_f32)
                w0[:ws[0].shape[0]] = ws[0]
                src = w0
            mp = torch.zeros(32, 128, device=dev, dtype=_f32)
            col = torch.where(row_reg < pe_flat.shape[1], pe_flat[row_half, row_reg.clamp(max=pe_flat.shape[1] - 1)], torch.full_like(row_reg, -2))
            live = col >= 0
            mp[m[live]] = src[:, col[live]].t()
            fp = groups(mp, hid)                                          # (16, 1, 64, 4)
            out.append(fp[:, 0].reshape(4, 4, 64, 4))                     # group t: the four float4 g = 0..3
    out.append(torch.zeros(2, 4, 64, 4, device=dev, dtype=_f32))
    stream = torch.cat(out, 0).contiguous()
    w_last = ws[6][0]
    fe = 20 * n_levels
    w_out = torch.zeros(2, 64 + 16 * tc, device=dev, dtype=_f32)
    for hh in range(2):
        feat = torch.tensor([32 * t + 8 * (r >> 2) + 4 * hh + (r & 3) for t in range(4) for r in range(16)], device=dev)
        w_out[hh, :64] = w_last[feat] / c
        tb = cond_flat[hh][:16 * tc]
        w_out[hh, 64:64 + tb.shape[0]] = torch.where(tb >= 0, w_last[(128 + tb).clamp(0, 127 + fe)], torch.zeros_like(tb, dtype=_f32))
    return stream, w_out


class SdfMlpPlan:
    """Weights of an SDFNetwork re-packed for gens_sdf_mlp.  Only the shipped architecture is supported
    (`supported(net)`); anything else keeps using the PyTorch layers on top of the K2 look-up kernels."""

    @staticmethod
    def supported(net):
        return (net.num_layers == 8 and tuple(net.skip_in) == (3,) and net.embed_fn_fine is not None and net.embed_fn_feat is not None
                and net.lin0.weight_v.shape == (128, 27) and net.init_feat_channels in (12, 20)
                and net.lin6.weight_v.shape[1] == 128 + 5 * net.init_feat_channels and net.lin2.weight_v.shape[0] == 101)

    @staticmethod
    def version(net):
        return tuple(p._version for p in net.parameters()) + tuple(p.data_ptr() for p in net.parameters())

    def __init__(self, net):
        assert SdfMlpPlan.supported(net), "gens_sdf_mlp is built for the architecture of confs/gens.conf:69-86"
        with torch.no_grad():
            ws, bs = [], []
            for l in range(7):
                lin = getattr(net, f"lin{l}")
                v, g = lin.weight_v.detach().to(_f32), lin.weight_g.detach().to(_f32)
                ws.append(v * (g / torch.linalg.norm(v, dim=1, keepdim=True)))
                bs.append(lin.bias.detach().to(_f32))
            dev = ws[0].device
            self.n_levels = net.init_feat_channels // 4
            self.wf, self.wb, self.bias = [], [], []
            c = 100.0 / math.log(2.0)      # pre-scaled forward streams (k6_sdfmlp.hip::softplus_t): hidden units travel as c * softplus
            for l in range(6):
                w = torch.zeros(128, ws[l].shape[1], device=dev, dtype=_f32)
                w[:ws[l].shape[0]] = ws[l]
                b = torch.zeros(128, device=dev, dtype=_f32)
                b[:bs[l].shape[0]] = bs[l]
                wbias = torch.cat([w, b[:, None]], 1)                              # bias = extra reduction row K_l (constant-1 input column)
                hidden = 0 if l == 0 else (101 if l == 3 else 128)                  # leading columns fed by (scaled) hidden units
                wbias[:, hidden:] *= c                                              # point encoding / skip columns / volume features / bias
                self.wf.append(_pack_b_groups(wbias))
                self.wb.append(_pack_b_groups(w.t().contiguous()))
                self.bias.append(b)
            # (the kernels' max / median activations drop NaNs: non-finite WEIGHTS are answered with NaN outputs, as the reference's layers would)
            self.finite = bool(torch.stack([torch.isfinite(w).all() for w in ws] + [torch.isfinite(b).all() for b in bs]).all())
            self.w_last = _c(ws[6][0].clone())
            self.w_last_scaled = self.w_last.clone()
            self.w_last_scaled[:128] /= c
            self.b_last = float(bs[6][0])
            self.scale = float(net.scale)
            self.value_stream, self.value_row = _pack_value_stream(ws, bs, self.n_levels)
            self.grad_stream, self.grad_row = _pack_grad_stream(ws, bs, self.n_levels)
            assert self.grad_stream.shape[0] == L.load().gens_sdf_grad_groups(self.n_levels) + 2
            self.value_units, self.value_w_out, vmax = _pack_value_units(ws, bs, self.n_levels)
            self.value_ok = vmax < 6.0e4
            self.overflow = torch.zeros(1, device=dev, dtype=torch.int32)
        self.wf_table, self.wb_table, self.bias_table = L.ptr_table(self.wf), L.ptr_table(self.wb), L.ptr_table(self.bias)
        self.key = SdfMlpPlan.version(net)

    def overflowed(self):
        """True if any split-half launch since the last call met a value outside the half range (synchronises)."""
        hit = bool(self.overflow.item())
        if hit:
            self.overflow.zero_()
        return hit


def sdf_mlp(plan, volumes, pts, index=None, want_grad=False, sdf_out=None, grad_out=None, precision="f32", count=None):
    """sdf (and d sdf/dx) of pts[index] written to sdf_out[index] / grad_out[index] (fresh, densely indexed outputs if
    no buffers are given).  volumes: packed VolumeSet with 3 or 5 levels.  No autograd graph is built (inference).
    precision: "f32" (exact float32 MFMA) or "f16x2" (split-half operands, ~1e-6 relative; check plan.overflowed()).
    count: optional (1,) int32 device tensor from compact_valid(): only the first `count` entries of `index` are evaluated."""
    assert isinstance(volumes, VolumeSet) and volumes.layout == L.LAYOUT_PACKED and volumes.n == plan.n_levels
    pts = _c(pts.detach().reshape(-1, 3).to(_f32))
    n = pts.shape[0] if index is None else index.shape[0]
    if sdf_out is None:
        sdf_out = torch.empty(pts.shape[0], 1, device=pts.device, dtype=_f32)
    if want_grad and grad_out is None:
        grad_out = torch.empty(pts.shape[0], 3, device=pts.device, dtype=_f32)
    idx = None if index is None else _c(index.to(torch.int64))
    fe = 20 * plan.n_levels
    flops = 2 * (27 * 128 + (128 + fe) * (4 * 128 + 101 + 1)) * (2 if want_grad else 1)
    nbytes = n * (12 + (16 if want_grad else 4) + (8 if idx is not None else 0))
    tag = ":grad" if want_grad else ":value"     # profile key: the two device kernels (sdf_mlp_k<FE, true / false>) are priced separately
    if isinstance(plan, SdfTrainStep):           # this training step's streams (gens_sdf_train_pack): same layout, bias on the device
        L.call("gens_sdf_mlp_dev", volumes.table, volumes.dim_table, volumes.n, plan.wf_table, plan.wb_table, L.ptr(plan.w_last),
               L.ptr(plan.b_last), 1.0, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(sdf_out),
               L.ptr(grad_out) if want_grad else None, L.stream(), nbytes=nbytes, flops=n * flops, live=None if count is None else (count, n),
               label="gens_sdf_mlp" + tag)
        return (sdf_out, grad_out) if want_grad else sdf_out
    if want_grad and kernels.sdf_grad == "transposed":
        # (also under "f16x2": the value + gradient pass stays float32 -- the split-half arithmetic covers the value-only passes)
        L.call("gens_sdf_grad", volumes.table, volumes.dim_table, volumes.n, L.ptr(plan.grad_stream), L.ptr(plan.grad_row), plan.b_last,
               plan.scale, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(sdf_out), L.ptr(grad_out),
               L.ptr(sdf_grad_stash(pts.device), torch.uint8), L.stream(),
               nbytes=nbytes, flops=n * flops, live=None if count is None else (count, n), label="gens_sdf_grad")
    elif precision == "f16x2" and not want_grad and plan.value_ok:
        L.call("gens_sdf_value_f16", volumes.table, volumes.dim_table, volumes.n, L.ptr(plan.value_units, torch.float16), L.ptr(plan.value_w_out),
               plan.b_last, plan.scale, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(sdf_out),
               L.ptr(plan.overflow, torch.int32), L.stream(), nbytes=nbytes, flops=n * flops, live=None if count is None else (count, n),
               label="gens_sdf_value_f16")
    elif not want_grad and kernels.sdf_value == "transposed":
        L.call("gens_sdf_value", volumes.table, volumes.dim_table, volumes.n, L.ptr(plan.value_stream), L.ptr(plan.value_row), plan.b_last,
               plan.scale, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(sdf_out), L.stream(), nbytes=nbytes,
               flops=n * flops, live=None if count is None else (count, n), label="gens_sdf_value")
    else:
        L.call("gens_sdf_mlp", volumes.table, volumes.dim_table, volumes.n, plan.wf_table, plan.wb_table, L.ptr(plan.w_last),
               L.ptr(plan.w_last_scaled), plan.b_last, plan.scale, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(sdf_out),
               L.ptr(grad_out) if want_grad else None, L.stream(), nbytes=nbytes, flops=n * flops, live=None if count is None else (count, n),
               label="gens_sdf_mlp" + tag)
    if not getattr(plan, "finite", True):
        _poison(idx, count, sdf_out, grad_out if want_grad else None)
    return (sdf_out, grad_out) if want_grad else sdf_out


def _poison(idx, count, *outs):
    """NaN into the evaluated rows of `outs` (index map / device-side count as the kernels take them): a network with non-finite weights."""
    for out in outs:
        if out is None:
            continue
        if idx is None:
            out.fill_(float("nan"))
        else:
            live = idx if count is None else idx[torch.arange(idx.shape[0], device=idx.device) < count.to(idx.device)[0]]
            out[live] = float("nan")


# ------------------------------------------------------------------------------------------------------------------
# K17  the SDF network of a training step: value, gradient, `smooth`, and the loss backward   (sdf_network.py:98-154)
# ------------------------------------------------------------------------------------------------------------------
class SdfTrainStep:
    """One training / fine-tune step's view of the SDF network (gens_sdf_train_*): the effective (weight-normed) matrices are packed
    into MFMA B streams ONCE, then any number of point batches are evaluated against them.

        step = SdfTrainStep(weights, biases, volumes, packed)      # weights[l] (out_l, in_l) with autograd history, l = 0..6
        y, g, s = step(pts)                                         # (N,1), (N,3), (N,3); differentiable once more (loss.backward())
        g0 = step.first_order(pts0)                                 # d sdf / dx only, no graph (implicit_surface.py:305-310)

    `volumes`: the planar (1,4,X,Y,Z) tensors the gradient goes to; `packed`: their (X,Y,Z,4) texel copy the kernels read."""

    @staticmethod
    def supported(net, n_levels):
        return SdfMlpPlan.supported(net) and float(net.scale) == 1.0 and n_levels in (3, 5) and net.init_feat_channels == 4 * n_levels

    def __init__(self, weights, biases, volumes, packed, raw=None, tv_masks=None):
        """weights / biases: the EFFECTIVE matrices with autograd history (torch._weight_norm's outputs) -- or, with raw = (weight_v list,
        weight_g list, bias list) of lin0..lin6, None: weight norm then happens inside the pack launch and its backward inside the
        gradient launch (gens_sdf_train_pack_wn / gens_sdf_train_wgrad), and the raw parameters are the autograd inputs.
        tv_masks: the mask pyramid; with it `step(pts, sel, tv=True)` also returns tv_regularization(volumes, masks) (implicit_surface.py:
        135-150) so that the dense TV gradient and the scattered look-up gradient of the volumes are formed in ONE buffer."""
        assert isinstance(packed, VolumeSet) and packed.layout == L.LAYOUT_PACKED and packed.n in (3, 5)
        self.volumes, self.packed = list(volumes), packed
        self.n_levels = packed.n
        self.raw = raw
        self.tv_masks = None if tv_masks is None else [_c(m.detach()) for m in tv_masks]
        dev = self.volumes[0].device if self.volumes else packed.tensors[0].device
        kin = 128 + 20 * self.n_levels
        self.kp = (kin + 1 + 7) // 8 * 8
        gf = [(27 + 1 + 7) // 8] + [self.kp // 8] * 5
        ntb = [1] + [(kin + 31) // 32] * 5
        self.wf = [torch.empty(4 * g * 64 * 4, device=dev, dtype=_f32) for g in gf]
        self.wb = [torch.empty(nt * 16 * 64 * 4, device=dev, dtype=_f32) for nt in ntb]
        self.wf_table, self.wb_table = L.ptr_table(self.wf), L.ptr_table(self.wb)
        with torch.no_grad():
            if raw is None:
                assert len(weights) == 7 and len(biases) == 7
                self.tensors = [*weights, *biases]
                w = [_c(t.detach().to(_f32)) for t in weights[:6]]
                b = [_c(t.detach().to(_f32)) for t in biases[:6]]
                assert tuple(w[0].shape) == (128, 27) and tuple(w[2].shape) == (101, kin) and tuple(w[5].shape) == (128, kin)
                L.call("gens_sdf_train_pack", L.ptr_table(w), L.ptr_table(b), self.n_levels, self.wf_table, self.wb_table, L.stream())
                self.w_last = _c(weights[6].detach().to(_f32)[0].clone())
                self.b_last = _c(biases[6].detach().to(_f32)[:1].clone())
            else:
                vs, gs, bs = raw
                assert len(vs) == len(gs) == len(bs) == 7
                self.tensors = [*vs, *gs, *bs]
                self.v = [_c(t.detach().to(_f32)) for t in vs]
                self.g = [_c(t.detach().to(_f32).reshape(-1)) for t in gs]
                b = [_c(t.detach().to(_f32)) for t in bs]
                assert tuple(self.v[0].shape) == (128, 27) and tuple(self.v[2].shape) == (101, kin) and tuple(self.v[6].shape) == (129, kin)
                self.scale = [torch.empty(t.shape[0], device=dev, dtype=_f32) for t in self.v]
                self.w_last, self.b_last = torch.empty(kin, device=dev, dtype=_f32), torch.empty(1, device=dev, dtype=_f32)
                L.call("gens_sdf_train_pack_wn", L.ptr_table(self.v), L.ptr_table(self.g), L.ptr_table(b), self.n_levels, L.ptr_table(self.scale),
                       self.wf_table, self.wb_table, L.ptr(self.w_last), L.ptr(self.b_last), L.stream(), label="gens_sdf_train_pack")

    def _forward(self, pts, sel=None):
        """sel (StepPoints): evaluate pts[sel.idx[:count]] with the count left on the device and write rows sel.idx[i] of sel's dense
        outputs (their other rows already hold the reference's defaults); None: every row of pts, fresh outputs."""
        n = pts.shape[0]
        dev = pts.device
        stash = torch.empty(L.load().gens_sdf_train_stash_bytes(n, 0), device=dev, dtype=torch.uint8)
        if sel is None:
            y, g, s = (torch.empty(n, k, device=dev, dtype=_f32) for k in (1, 3, 3))
            idx = cnt = None
        else:
            y, g, s, idx, cnt = sel.y, sel.g, sel.s, sel.idx, sel.counts[0:1]
        fe = 20 * self.n_levels
        flops = 4 * 2 * (27 * 128 + (128 + fe) * (4 * 128 + 101 + 1))
        L.call("gens_sdf_train_fwd", self.packed.table, self.packed.dim_table, self.n_levels, self.wf_table, self.wb_table, L.ptr(self.w_last),
               L.ptr(self.b_last), L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32), L.ptr(stash, torch.uint8), L.ptr(y), L.ptr(g),
               L.ptr(s), L.stream(), nbytes=n * 40, flops=n * flops, live=None if cnt is None else (cnt, n))
        return y, g, s

    def __call__(self, pts, sel=None, tv=False):
        """-> (y, g, s) [, tv_reg when tv=True (needs tv_masks)]."""
        out = _SdfTrain.apply(_c(pts.detach().reshape(-1, 3).to(_f32)), self, sel, bool(tv), *self.tensors, *self.volumes)
        return out if tv else out[:3]

    @torch.no_grad()
    def first_order(self, pts):
        return self._forward(_c(pts.detach().reshape(-1, 3).to(_f32)))[1]


class _SdfTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, step, sel, tv, *tensors):
        ctx.step, ctx.sel = step, sel
        ctx.shapes = [t.shape for t in tensors]
        ctx.n_par = len(step.tensors)
        y, g, s = step._forward(pts, sel)
        tv_out = None
        if tv:
            assert step.tv_masks is not None, "SdfTrainStep(tv_masks=...) is needed for tv=True"
            nl = step.n_levels
            vols = [_c(v.detach()) for v in step.volumes]
            ctx.tv_dims = [d for v in vols for d in v.shape[-3:]]
            partial = torch.empty(L.load().gens_tv_levels_blocks(L.int_table(ctx.tv_dims), nl), 4, device=pts.device, dtype=_f32)
            tv_out = torch.empty(1 + nl, device=pts.device, dtype=_f32)
            L.call("gens_tv_levels_fwd", L.ptr_table(vols, align=16), L.ptr_table(step.tv_masks, align=16), L.int_table(ctx.tv_dims), nl, L.ptr(partial),
                   L.ptr(tv_out), L.stream(), nbytes=sum(20 * v[0, 0].numel() for v in vols))
            ctx.save_for_backward(pts, tv_out, *vols)
            return y, g, s, tv_out[0]
        ctx.save_for_backward(pts)
        return y, g, s, None

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, y_bar, g_bar, s_bar, tv_bar):
        step, sel = ctx.step, ctx.sel
        pts, *tv_saved = ctx.saved_tensors
        n, dev = pts.shape[0], pts.device
        idx, cnt = (None, None) if sel is None else (sel.idx, sel.counts[0:1])
        nl = step.n_levels
        cf, fe, kin = 4 * nl, 20 * nl, 128 + 20 * nl
        fep = step.kp - 128
        npad = (n + 31) // 32 * 32
        f = lambda *shape: torch.empty(*shape, device=dev, dtype=_f32)  # noqa: E731
        lop, rh, re, r0 = f(npad, 4, 6, 128), f(5, npad, 4, 128), f(npad, 4, fep), f(npad, 4, 32)      # point-major operand rows
        f_hat, mu_f, lam_f, w6p = f(npad, cf), f(npad, cf), f(npad, cf), f(npad // 32, step.kp)
        stash = torch.empty(L.load().gens_sdf_train_stash_bytes(n, 1), device=dev, dtype=torch.uint8)
        cot = [None if t is None else _c(t.to(_f32)) for t in (y_bar, g_bar, s_bar)]
        flops = 8 * 2 * (27 * 128 + (128 + fe) * (4 * 128 + 101 + 1))
        live = None if cnt is None else (cnt, n)
        L.call("gens_sdf_train_bwd", step.packed.table, step.packed.dim_table, nl, step.wf_table, step.wb_table, L.ptr(step.w_last), L.ptr(pts),
               L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32), L.ptr(cot[0]), L.ptr(cot[1]), L.ptr(cot[2]), L.ptr(stash, torch.uint8), L.ptr(lop),
               L.ptr(rh), L.ptr(re), L.ptr(r0), L.ptr(f_hat), L.ptr(mu_f), L.ptr(lam_f), L.ptr(w6p), L.stream(),
               nbytes=n * (40 + 4 * (4 * 6 * 128 + 6 * 4 * 128 + 4 * fep + 4 * 32 + 3 * cf)), flops=n * flops, live=live)
        # weight gradients: eleven products over the 4 * npad operand rows in ONE launch (rows of padding points are zero on one side of
        # every product).  Per layer l = 1..5 the hidden columns (lop_l^T rh_l) and the conditioning columns + bias (lop_l^T re) are
        # neighbours in the unit list, so the second product finds lop_l's slab in L2 instead of reading it from HBM again; layer 0 last.
        k = 4 * npad
        fl = 4                                                            # bytes per float
        a_ptr, b_ptr, ldb, ms, ns = [], [], [], [], []
        for l in range(1, 6):
            a_ptr += [lop.data_ptr() + fl * 128 * l] * 2
            b_ptr += [rh.data_ptr() + fl * (l - 1) * k * 128, re.data_ptr()]
            ldb += [128, fep]
            ms += [128, 128]
            ns += [128, fep]
        a_ptr.append(lop.data_ptr())
        b_ptr.append(r0.data_ptr())
        ldb.append(32)
        ms.append(128)
        ns.append(32)
        n_prod = len(ms)
        mi, ni = L.int_table(ms), L.int_table(ns)
        ws = f(L.load().gens_gemm_tn_batch_workspace(n_prod, mi, ni, k))
        sizes = [m * n_ for m, n_ in zip(ms, ns)]
        cc = f(sum(sizes))
        tab = lambda v: C.cast((C.c_void_p * len(v))(*v), C.POINTER(C.c_void_p))  # noqa: E731
        if idx is None:
            L.call("gens_gemm_tn_batch", n_prod, tab(a_ptr), L.int_table([768] * n_prod), tab(b_ptr), L.int_table(ldb), mi, ni, k,
                   L.ptr(ws), L.ptr(cc), L.stream(), nbytes=fl * k * (768 + 32 + 5 * 128 + fep), flops=2 * k * sum(sizes))
        else:       # only the operand rows of the points that exist (32 points -> 128 rows per workgroup of the backward launch)
            L.call("gens_gemm_tn_batch_live", n_prod, tab(a_ptr), L.int_table([768] * n_prod), tab(b_ptr), L.int_table(ldb), mi, ni, k,
                   L.ptr(cnt, torch.int32), 32, 128, L.ptr(ws), L.ptr(cc), L.stream(), nbytes=fl * k * (768 + 32 + 5 * 128 + fep),
                   flops=2 * k * sum(sizes), live=live, label="gens_gemm_tn_batch")
        w6s = w6p.sum(0)
        n_par = ctx.n_par
        if step.raw is not None:
            # d loss / d (weight_v, weight_g, bias) of lin0..lin6 in ONE launch, weight norm's backward included; the 21 gradients are views
            # of one flat buffer (contiguous each: autograd installs them as .grad without a copy)
            sizes_v = [v.numel() for v in step.v]
            rows = [v.shape[0] for v in step.v]
            flat = f(sum(sizes_v) + 2 * sum(rows))
            dv, dg, db, off = [], [], [], 0
            for v in step.v:
                dv.append(flat[off:off + v.numel()].view(v.shape))
                off += v.numel()
            for r in rows:
                dg.append(flat[off:off + r])
                off += r
            for r in rows:
                db.append(flat[off:off + r])
                off += r
            L.call("gens_sdf_train_wgrad", L.ptr_table(step.v), L.ptr_table(step.g), nl, L.ptr(cc), L.ptr(w6s), L.ptr_table(dv), L.ptr_table(dg),
                   L.ptr_table(db), L.stream())
            g_par = [*dv, *[d.view(ctx.shapes[7 + k]) for k, d in enumerate(dg)], *db]
        else:
            parts, off = [], 0
            for m, n_ in zip(ms, ns):
                parts.append(cc[off:off + m * n_].view(m, n_))
                off += m * n_
            w0 = parts[10]
            g_w, g_b = [w0[:, :27]], [w0[:, 27]]
            for l in range(1, 6):
                rows = 101 if l == 2 else 128
                h, e_l = parts[2 * (l - 1)], parts[2 * (l - 1) + 1]
                g_w.append(torch.cat([h, e_l[:, :fe]], 1)[:rows])
                g_b.append(e_l[:rows, fe])
            w6 = torch.zeros(ctx.shapes[6], device=dev, dtype=_f32)
            w6[0] = w6s[:kin]
            b6 = torch.zeros(ctx.shapes[13], device=dev, dtype=_f32)
            b6[0] = w6s[kin]
            g_par = [*g_w, w6, *g_b, b6]
        # volume gradients: the dense TV gradient (when the step carries the regulariser) is WRITTEN first, the look-up's scatter adds into it
        g_vols = [None] * nl
        if any(ctx.needs_input_grad[4 + n_par:]):
            have_tv = bool(tv_saved) and tv_bar is not None
            if have_tv:
                tv_out, *vols = tv_saved
                g_vols = [torch.empty(s, device=dev, dtype=_f32) for s in ctx.shapes[n_par:]]
                L.call("gens_tv_levels_bwd", L.ptr_table(list(vols), align=16), L.ptr_table(step.tv_masks, align=16), L.int_table(ctx.tv_dims), nl,
                       L.ptr(tv_out), L.ptr(_c(tv_bar.detach().to(_f32).reshape(1))), L.ptr_table(g_vols, align=16), L.stream(),
                       nbytes=sum(36 * v[0, 0].numel() for v in vols))
            else:
                g_vols = [torch.zeros(s, device=dev, dtype=_f32) for s in ctx.shapes[n_par:]]
            L.call("gens_sdf_train_scatter", step.packed.dim_table, nl, L.ptr(pts), L.ptr(cot[1]), L.ptr(cot[2]), L.ptr(f_hat), L.ptr(mu_f),
                   L.ptr(lam_f), L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32), L.ptr_table(g_vols), L.stream(), nbytes=n * (36 + 3 * 4 * cf),
                   live=live)
        return (None, None, None, None, *g_par, *g_vols)


# ------------------------------------------------------------------------------------------------------------------
# K14  a^T b for tall operands: the weight-gradient product of the training step
# ------------------------------------------------------------------------------------------------------------------
MATMUL_TN_MIN_ROWS = 8192      # below this the library GEMM is as good


def _gemm_tn(a, b):
    a, b = _c(a.detach().to(_f32)), _c(b.detach().to(_f32))
    k, m = a.shape
    n = b.shape[1]
    slabs = L.load().gens_gemm_tn_slabs(k, m, n)
    ws = torch.empty(slabs * m * n, device=a.device, dtype=_f32)
    c = torch.empty(m, n, device=a.device, dtype=_f32)
    L.call("gens_gemm_tn", L.ptr(a), L.ptr(b), k, m, n, L.ptr(ws), L.ptr(c), L.stream(), nbytes=4 * (k * (m + n) + m * n), flops=2 * k * m * n)
    return c


class _MatmulTN(torch.autograd.Function):
    """c = a^T b.  Its derivatives are tall-times-small products (_MatmulNN), whose derivatives are again a^T b products: every
    reduction over the rows stays on K14 through the second and third derivatives the SDF network takes through its layers."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _gemm_tn(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return matmul_nn(b, g.t()) if ctx.needs_input_grad[0] else None, matmul_nn(a, g) if ctx.needs_input_grad[1] else None


class _MatmulNN(torch.autograd.Function):
    """y = x w for a tall x (K, M) and a small w (M, N): the library product, with d/dw = x^T g routed to K14."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return matmul_nn(g, w.t()) if ctx.needs_input_grad[0] else None, matmul_tn(x, g) if ctx.needs_input_grad[1] else None


def _tall(a, *others):
    return a.is_cuda and a.dim() == 2 and a.shape[0] >= MATMUL_TN_MIN_ROWS and all(t.dtype == _f32 for t in (a, *others))


def matmul_tn(a, b):
    """a (K, M), b (K, N) -> a^T b (M, N).  Tall float32 device operands go to gens_gemm_tn (K split over the chip, fp32 MFMA);
    anything else to torch."""
    if _tall(a, b) and a.shape[1] <= 1024 and b.shape[1] <= 1024:
        return _MatmulTN.apply(a, b)
    return a.t() @ b


def matmul_nn(x, w):
    """x (K, M) @ w (M, N): torch's product; for tall x the gradient w.r.t. w is a K14 product."""
    if _tall(x, w) and x.shape[1] <= 1024 and w.shape[1] <= 1024 and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        return _MatmulNN.apply(x, w)
    return x @ w


# ------------------------------------------------------------------------------------------------------------------
# K15  3 x 3 x 3 convolutions of the cost-volume U-Net (reg_network.py:7-50), forward / data gradient / weight gradient
# ------------------------------------------------------------------------------------------------------------------
def _conv_pad_last(t, block):
    n = t.shape[-1]
    m = (n + block - 1) // block * block
    return _c(t) if m == n else _c(torch.nn.functional.pad(t, (0, m - n)))


def _conv_gather(q, w_abt, bias, stride, reverse=False):
    """P = gather(Q): q (cq, sX, sY, sZ).  w_abt (A, B, 27): output channels A over input channels B -- or, with `reverse`, the
    stride-1 scatter through the same taps: output channels B over input channels A, taps reversed."""
    wt = w_abt.flip(-1).permute(0, 2, 1) if reverse else w_abt.permute(1, 2, 0)          # (in, 27, out)
    cq, cp = wt.shape[0], wt.shape[2]
    assert q.shape[0] == cq and all(d % stride == 0 for d in q.shape[1:])
    dims = [d // stride for d in q.shape[1:]]
    wt = _conv_pad_last(wt, 8 if cp > 4 else 4)
    p = torch.empty(cp, *dims, device=q.device, dtype=_f32)
    n = p[0].numel()
    L.call("gens_conv3d_gather", L.ptr(q, align=16), L.ptr(wt), L.ptr(None if bias is None else _c(bias.detach().to(_f32))), cp, cq, L.int_table(dims), stride,
           L.ptr(p), L.stream(), nbytes=4 * (q.numel() + p.numel()), flops=2 * 27 * cp * cq * n)
    return p


def _conv_scatter2(p, w_abt):
    """Q = scatter(P), stride 2: p (cp, X, Y, Z), w_abt (cp, cq, 27) -> q (cq, 2X, 2Y, 2Z)."""
    cp, cq = w_abt.shape[:2]
    assert p.shape[0] == cp
    dims = list(p.shape[1:])
    wt = _conv_pad_last(w_abt.permute(0, 2, 1), 8 if cq > 4 else 4)                      # (cp, 27, cq padded)
    q = torch.empty(cq, *[2 * d for d in dims], device=p.device, dtype=_f32)
    L.call("gens_conv3d_scatter2", L.ptr(p, align=16), L.ptr(wt), cp, cq, L.int_table(dims), L.ptr(q), L.stream(),
           nbytes=4 * (q.numel() + p.numel()), flops=2 * 27 * cp * cq * p[0].numel())
    return q


def _conv_wgrad(p, q, stride):
    """dW (cp, cq, 27) = sum over the voxels o of P of P[a][o] Q[b][s o + t - 1]."""
    cp, cq = p.shape[0], q.shape[0]
    dims = L.int_table(p.shape[1:])
    parts = L.load().gens_conv3d_wgrad_parts(cp, cq, dims)
    cpp, cqp = (cp + 3) // 4 * 4, (cq + 7) // 8 * 8
    ws = torch.empty(parts, cpp, cqp, 27, device=p.device, dtype=_f32)
    L.call("gens_conv3d_wgrad", L.ptr(p), L.ptr(q), cp, cq, dims, stride, L.ptr(ws), L.stream(),
           nbytes=4 * (q.numel() + p.numel()), flops=2 * 27 * cp * cq * p[0].numel())
    return ws.sum(0)[:cp, :cq]


def _plane_sums(x2):
    """Per-row sums of a (c, n) float32 tensor through K16's statistics pass (float64 accumulation, the whole chip per row)."""
    c, n = x2.shape
    part = torch.empty(c, L.load().gens_instnorm_blocks(c, n), 2, device=x2.device, dtype=torch.float64)
    L.call("gens_instnorm_stats", L.ptr(x2, align=16), c, n, L.ptr(part, torch.float64), L.stream(), nbytes=4 * c * n)
    return part[:, :, 0].sum(1).to(_f32)


class _Conv3d(torch.autograd.Function):
    """torch.nn.functional.conv3d(x, w, b, stride, padding=1) for a 3 x 3 x 3 kernel and batch 1."""

    @staticmethod
    def forward(ctx, x, w, b, stride):
        x3 = aligned16(x.detach()[0].to(_f32))
        ctx.save_for_backward(x3, w)
        ctx.stride, ctx.has_bias = stride, b is not None
        return _conv_gather(x3, w.detach().to(_f32).reshape(w.shape[0], w.shape[1], 27), b, stride)[None]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x3, w = ctx.saved_tensors
        g3 = _c(gy[0].to(_f32))
        w3 = w.detach().to(_f32).reshape(w.shape[0], w.shape[1], 27)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (_conv_gather(g3, w3, None, 1, reverse=True) if ctx.stride == 1 else _conv_scatter2(g3, w3))[None]
        if ctx.needs_input_grad[1]:
            gw = _conv_wgrad(g3, x3, ctx.stride).reshape(w.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _plane_sums(g3.reshape(g3.shape[0], -1))                # (aten::sum over a (4, 256^3) tensor takes 3.6 ms: 4 outputs, no parallelism)
        return gx, gw, gb, None


class _ConvTranspose3d(torch.autograd.Function):
    """torch.nn.functional.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1) for a 3 x 3 x 3 kernel, batch 1, no bias."""

    @staticmethod
    def forward(ctx, x, w):
        x3 = aligned16(x.detach()[0].to(_f32))
        ctx.save_for_backward(x3, w)
        return _conv_scatter2(x3, w.detach().to(_f32).reshape(w.shape[0], w.shape[1], 27))[None]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x3, w = ctx.saved_tensors
        g3 = _c(gy[0].to(_f32))
        w3 = w.detach().to(_f32).reshape(w.shape[0], w.shape[1], 27)
        gx = _conv_gather(g3, w3, None, 2)[None] if ctx.needs_input_grad[0] else None
        gw = _conv_wgrad(x3, g3, 2).reshape(w.shape) if ctx.needs_input_grad[1] else None
        return gx, gw


def conv3d(x, weight, bias=None, stride=1):
    """3 x 3 x 3 convolution, padding 1, stride 1 or 2: x (1, cin, X, Y, Z) -> (1, cout, X / s, Y / s, Z / s)."""
    assert x.dim() == 5 and x.shape[0] == 1 and tuple(weight.shape[2:]) == (3, 3, 3) and weight.shape[1] == x.shape[1] and stride in (1, 2)
    return _Conv3d.apply(x, weight, bias, stride)


def conv_transpose3d(x, weight):
    """3 x 3 x 3 transposed convolution, stride 2, padding 1, output_padding 1: x (1, cin, X, Y, Z) -> (1, cout, 2X, 2Y, 2Z)."""
    assert x.dim() == 5 and x.shape[0] == 1 and tuple(weight.shape[2:]) == (3, 3, 3) and weight.shape[0] == x.shape[1]
    return _ConvTranspose3d.apply(x, weight)


# ------------------------------------------------------------------------------------------------------------------
# K16  InstanceNorm3d (no affine) + ReLU of the U-Net blocks (reg_network.py:16-17,39-40)
# ------------------------------------------------------------------------------------------------------------------
_f64 = torch.float64


class _InstNormRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, skip=None):
        x2 = aligned16(x.detach().to(_f32)).reshape(x.shape[1], -1)
        c, n = x2.shape
        blocks = L.load().gens_instnorm_blocks(c, n)
        part = torch.empty(c, blocks, 2, device=x.device, dtype=_f64)
        L.call("gens_instnorm_stats", L.ptr(x2, align=16), c, n, L.ptr(part, _f64), L.stream(), nbytes=4 * c * n)
        s = part.sum(1) / n                                                        # float64: mean, mean of squares
        mean = s[:, 0]
        mr = torch.stack([mean, torch.rsqrt((s[:, 1] - mean * mean).clamp_min(0.0) + eps)], 1).to(_f32)
        y = torch.empty_like(x2)
        if skip is None:
            L.call("gens_instnorm_relu_fwd", L.ptr(x2), L.ptr(mr), c, n, L.ptr(y), L.stream(), nbytes=8 * c * n)
        else:
            assert skip.shape == x.shape
            L.call("gens_instnorm_relu_add_fwd", L.ptr(x2), L.ptr(mr), L.ptr(_c(skip.detach().to(_f32))), c, n, L.ptr(y), L.stream(), nbytes=12 * c * n)
        ctx.save_for_backward(x2, mr)
        ctx.blocks = blocks
        return y.reshape(x.shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x2, mr = ctx.saved_tensors
        c, n = x2.shape
        g2 = _c(gy.to(_f32)).reshape(c, n)
        part = torch.empty(c, ctx.blocks, 2, device=x2.device, dtype=_f64)
        L.call("gens_instnorm_relu_bwd_stats", L.ptr(x2), L.ptr(g2), L.ptr(mr), c, n, L.ptr(part, _f64), L.stream(), nbytes=8 * c * n)
        m12 = (part.sum(1) / n).to(_f32)
        gx = torch.empty_like(x2)
        L.call("gens_instnorm_relu_bwd", L.ptr(x2), L.ptr(g2), L.ptr(mr), L.ptr(m12), c, n, L.ptr(gx), L.stream(), nbytes=12 * c * n)
        return gx.reshape(gy.shape), None, (gy if ctx.needs_input_grad[2] else None)


def instnorm_relu(x, eps=1e-5, skip=None):
    """relu(instance_norm(x)) [+ skip] for x (1, c, ...): per-channel statistics over the plane, biased variance, no affine parameters."""
    assert x.shape[0] == 1 and x.dim() >= 3
    return _InstNormRelu.apply(x, float(eps), skip)


# ------------------------------------------------------------------------------------------------------------------
# K7  fused source-view look-up + BlendingNetwork (inference)   (projector.py:278-349 + blending_network.py:69-118)
# ------------------------------------------------------------------------------------------------------------------
def _pad32(b):
    out = torch.zeros(32 * ((b.numel() + 31) // 32), device=b.device, dtype=_f32)
    out[:b.numel()] = b.reshape(-1)
    return out


def _pack_blend_t(layers, n_feat):
    """Weight stream and tables of gens_blend_views_t (k7t_blend.hip).  Activations live in "quad layout" (feature f = 4 kq + q: register
    kq of lane group q); an A fragment of (M tile T, group g of four K quads) holds for lane (m, qk) and j = 0..3 the weight
    W[16 T + 4 (m & 3) + (m >> 2)][slot 16 g + 4 j + qk], so that accumulator register i of lane group q is output feature 4 (4 T + i) + q.
    `layers`: dict name -> (weight, bias).  Returns (stream (G + 2, 64, 4), tab (entries, 4, 8)); the two trailing groups are zero."""
    f = n_feat
    xq = (f + 1) // 4
    dev = layers["rd1"][0].device
    lane = torch.arange(64, device=dev)
    m, qk = lane & 15, lane >> 4
    row_in_tile = 4 * (m & 3) + (m >> 2)

    def product(w, slots, m_tiles, bias=None):
        """w (O, I); slots: list of source columns per input slot (-1: the bias, -2: nothing), padded to whole quads."""
        o = w.shape[0]
        aug = torch.cat([w, (bias if bias is not None else torch.zeros(o, device=dev))[:, None], torch.zeros(o, 1, device=dev)], 1)
        aug = torch.cat([aug, torch.zeros(16 * m_tiles - o, aug.shape[1], device=dev)], 0) if 16 * m_tiles > o else aug
        cols = torch.tensor([c if c >= 0 else (w.shape[1] if c == -1 else w.shape[1] + 1) for c in slots], device=dev)
        nq = (len(slots) + 3) // 4
        cols = torch.cat([cols, torch.full((4 * nq - len(slots),), w.shape[1] + 1, device=dev)])
        groups = []
        for t in range(m_tiles):
            rows = 16 * t + row_in_tile
            for g in range((nq + 3) // 4):
                frag = torch.zeros(64, 4, device=dev, dtype=_f32)
                for j in range(4):
                    kq = 4 * g + j
                    if kq < nq:
                        frag[:, j] = aug[rows, cols[4 * kq + qk]]
                groups.append(frag)
        return groups

    rd1, rd2, b1, b2, v1, v2, u1, u2, r1, r2, r3 = (layers[k] for k in ("rd1", "rd2", "b1", "b2", "v1", "v2", "u1", "u2", "r1", "r2", "r3"))
    xt = (xq + 3) // 4
    g = []
    g += product(rd1[0], [0, 1, 2, 3], 1)
    g += product(rd2[0], list(range(16)), xt)
    pad = [-2] * (4 * xq - f)
    g += product(b1[0], list(range(f)) + pad + list(range(f, 2 * f)) + pad, 4)                         # mean | var, once per point
    g += product(b1[0], list(range(2 * f, 3 * f)) + [-1], 4, b1[1])                                    # x and the bias (slot F)
    g += product(b2[0], list(range(64)), 2)
    g += product(v1[0], list(range(32)), 2)
    g += product(v2[0][:32], list(range(32)), 2)
    g += product(u1[0], list(range(32)), 2)
    g += product(r1[0], list(range(36)) + [36, -1, -2, -2], 1, r1[1])
    g += product(r2[0], list(range(16)), 1)
    stream = torch.stack(g + [torch.zeros(64, 4, device=dev, dtype=_f32)] * 2).contiguous()

    def acc_bias(b, m_tiles):            # [q][4 T + i] = b[16 T + 4 i + q]
        full = torch.zeros(32, device=dev, dtype=_f32)
        full[:min(b.shape[0], 16 * m_tiles)] = b[:16 * m_tiles]
        out = torch.zeros(4, 8, device=dev, dtype=_f32)
        for q in range(4):
            for t in range(m_tiles):
                for i in range(4):
                    out[q, 4 * t + i] = full[16 * t + 4 * i + q]
        return out

    def dot_row(w):                      # [q][kq] = w[4 kq + q]
        full = torch.zeros(32, device=dev, dtype=_f32)
        full[:w.shape[0]] = w
        return full.reshape(8, 4).t().contiguous()

    tab = torch.stack([acc_bias(rd1[1], 1), acc_bias(rd2[1], xt), acc_bias(b2[1], 2), acc_bias(v1[1], 2), acc_bias(v2[1][:32], 2),
                       acc_bias(u1[1], 2), acc_bias(r2[1], 1), dot_row(v2[0][32]), dot_row(u2[0][0]), dot_row(r3[0][0])]).contiguous()
    return stream, tab


class BlendPlan:
    """Weights of a BlendingNetwork re-packed for gens_blend_views (anti_alias_pooling=True, d_feature <= 20)."""

    @staticmethod
    def supported(net):
        return bool(getattr(net, "anti_alias_pooling", False)) and net.base_fc[0].weight.shape[0] == 64 and net.rgb_fc[0].weight.shape[1] == 37

    @staticmethod
    def version(net):
        return tuple(p._version for p in net.parameters()) + tuple(p.data_ptr() for p in net.parameters())

    def __init__(self, net):
        assert BlendPlan.supported(net)
        g = lambda m: (m.weight.detach().to(_f32), m.bias.detach().to(_f32))  # noqa: E731
        with torch.no_grad():
            rd1, rd2 = g(net.ray_dir_fc[0]), g(net.ray_dir_fc[2])
            b1, b2 = g(net.base_fc[0]), g(net.base_fc[2])
            v1, v2 = g(net.vis_fc[0]), g(net.vis_fc[2])
            u1, u2 = g(net.vis_fc2[0]), g(net.vis_fc2[2])
            r1, r2, r3 = g(net.rgb_fc[0]), g(net.rgb_fc[2]), g(net.rgb_fc[4])
            self.n_feat = rd2[0].shape[0]                  # 3 + d_feature
            P = _pack_b_groups      # grouped B streams: one global_load_dwordx4 per 4 MFMAs (layout in k7_blend.hip)
            N = _pack_b16           # narrow layers (<= 16 outputs): 16x16x4 tiles, reduction groups (k0, S)
            self.tensors = [N(rd1[0], [(0, 2)]), _pad32(rd1[1]), P(rd2[0]), _pad32(rd2[1]), P(b1[0]), _pad32(b1[1]), P(b2[0]), _pad32(b2[1]),
                            P(v1[0]), _pad32(v1[1]), P(v2[0][:32]), _pad32(v2[1][:32]), _c(v2[0][32].clone()),
                            P(u1[0]), _pad32(u1[1]), _c(u2[0][0].clone()),
                            N(r1[0], [(0, 4), (16, 4), (32, 2)]), _pad32(r1[1]), N(r2[0], [(0, 4)]), _pad32(r2[1]), _c(r3[0][0].clone())]
            self.scalars = (C.c_float * 4)(float(v2[1][32]), float(u2[1][0]), float(r3[1][0]), float(net.s.detach().abs()))
            self.finite = bool(torch.stack([torch.isfinite(p.detach()).all() for p in net.parameters()]).all())
            self.t_stream, self.t_tab = _pack_blend_t(dict(rd1=rd1, rd2=rd2, b1=b1, b2=b2, v1=v1, v2=v2, u1=u1, u2=u2, r1=r1, r2=r2, r3=r3),
                                                      self.n_feat)
            assert self.t_stream.shape[0] == L.load().gens_blend_views_t_groups((self.n_feat - 3) // 4) + 2
        self.table = L.ptr_table(self.tensors)
        self.key = BlendPlan.version(net)


def blend_views(plan, views, pts, index=None, rgb_out=None, vis_out=None, count=None):
    """Blended colour of pts[index] (N,3) and the per-source in-frustum flags (N,S) written at index (dense outputs)."""
    pts = _c(pts.detach().reshape(-1, 3).to(_f32))
    n = pts.shape[0] if index is None else index.shape[0]
    s = views.nv - 1
    nl = len(views.feat_tex)
    assert plan.n_feat == 3 + 4 * nl, "colour network width does not match the feature pyramid"
    if rgb_out is None:
        rgb_out = torch.zeros(pts.shape[0], 3, device=pts.device, dtype=_f32)
    if vis_out is None:
        vis_out = torch.zeros(pts.shape[0], s, device=pts.device, dtype=torch.uint8)
    idx = None if index is None else _c(index.to(torch.int64))
    hw = [d for f in views.feat_tex for d in f.shape[1:3]]
    feats = [aligned16(f.detach()) for f in views.feat_tex]
    f = plan.n_feat
    flops = 2 * s * (4 * 16 + 16 * f + 3 * f * 64 + 64 * 32 + 32 * 32 + 32 * 33 + 32 * 32 + 32 + 37 * 16 + 16 * 8 + 8)
    nbytes = n * (12 + 12 + s + (8 if idx is not None else 0))
    if 2 <= s <= 4 and kernels.blend == "transposed":                  # two to four source views: the transposed kernel (k7t_blend.hip)
        L.call("gens_blend_views_t", L.ptr_table(feats, align=16), L.int_table(hw), nl, L.ptr(aligned16(views.imgs_tex.detach()), align=16),
               L.ptr(views.w2c), L.ptr(views.intr), L.ptr(views.c2w), views.nv, L.ptr(plan.t_stream), L.ptr(plan.t_tab), plan.scalars, L.ptr(pts),
               L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32), L.ptr(rgb_out), L.ptr(vis_out, torch.uint8), L.stream(),
               live=None if count is None else (count, n), nbytes=nbytes, flops=n * flops, label="gens_blend_views")
        if not plan.finite:
            _poison(idx, count, rgb_out)
        return rgb_out, vis_out
    L.call("gens_blend_views", L.ptr_table(feats, align=16), L.int_table(hw), nl, L.ptr(aligned16(views.imgs_tex.detach()), align=16), L.ptr(views.w2c), L.ptr(views.intr),
           L.ptr(views.c2w), views.nv, plan.table, plan.scalars, L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(count, torch.int32),
           L.ptr(rgb_out), L.ptr(vis_out, torch.uint8), L.stream(), live=None if count is None else (count, n), nbytes=nbytes, flops=n * flops)
    if not plan.finite:
        _poison(idx, count, rgb_out)
    return rgb_out, vis_out


# ------------------------------------------------------------------------------------------------------------------
# K18  lookup_feature + BlendingNetwork of a training step, forward and backward   (projector.py:278-349, blending_network.py:69-118)
# ------------------------------------------------------------------------------------------------------------------
def blend_params(net):
    """The 23 raw parameters of a BlendingNetwork in the order gens_blend_train_* read them."""
    mods = [net.ray_dir_fc[0], net.ray_dir_fc[2], net.base_fc[0], net.base_fc[2], net.vis_fc[0], net.vis_fc[2], net.vis_fc2[0], net.vis_fc2[2],
            net.rgb_fc[0], net.rgb_fc[2], net.rgb_fc[4]]
    out = []
    for m in mods:
        out += [m.weight, m.bias]
    return out + [net.s]


class _BlendTrain(torch.autograd.Function):
    """(pts, views, *23 parameters, imgs_tex, *feat_tex) -> (rgb (N,3), vis (N,S) uint8).  Backward: one launch that recomputes the forward
    of its rows and walks the layers in reverse, one batched K14 launch for the eleven [dW | db], K4's backward for the maps."""

    @staticmethod
    def forward(ctx, pts, views, sel, *tensors):
        params, imgs_tex, feat_tex = tensors[:23], tensors[23], tensors[24:]
        # sel (StepPoints): the ray samples among sel.idx[:count_ray] (the list is sorted, ray samples first), dense outputs in sel
        n, s, nl = (pts.shape[0] if sel is None else sel.n_ray), views.nv - 1, len(feat_tex)
        dev = pts.device
        idx, cnt = (None, None) if sel is None else (sel.idx, sel.counts[1:2])
        w = [_c(p.detach().to(_f32)).reshape(-1) if p.dim() == 0 else _c(p.detach().to(_f32)) for p in params]
        feats = [aligned16(f.detach()) for f in feat_tex]
        imgs = aligned16(imgs_tex.detach())
        hw = [d for f in feats for d in f.shape[1:3]]
        if sel is None:
            rgb = torch.empty(n, 3, device=dev, dtype=_f32)
            vis = torch.empty(n, s, device=dev, dtype=torch.uint8)
        else:
            rgb, vis = sel.rgb, sel.vis
        f = 3 + 4 * nl
        flops = 2 * s * (4 * 16 + 16 * f + 3 * f * 64 + 64 * 32 + 32 * 32 + 32 * 33 + 32 * 32 + 32 + 37 * 16 + 16 * 8 + 8)
        ctx.args = (L.ptr_table(feats, align=16), L.int_table(hw), nl, L.ptr(imgs, align=16), L.ptr(views.w2c), L.ptr(views.intr), L.ptr(views.c2w),
                    views.nv, L.ptr_table(w), L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32))
        ctx.keep = (feats, imgs, w, views, pts, hw, idx, cnt)
        ctx.meta = (n, s, nl, f, flops, [p.shape for p in params], [t.shape for t in feat_tex], imgs_tex.shape)
        ctx.live = None if cnt is None else (cnt, n)
        if 2 <= s <= 4 and kernels.blend_train_fwd == "transposed":
            # the forward values come from the TRANSPOSED inference kernel (k7t_blend.hip: 5 - 6 x the rate of the row-major training kernel),
            # its weight stream packed from this step's raw parameters by one launch; the backward launch recomputes what it differentiates
            groups = L.load().gens_blend_views_t_groups(nl)
            wstream = torch.empty((groups + 2) * 64 * 4, device=dev, dtype=_f32)
            tab, sc = torch.empty(320, device=dev, dtype=_f32), torch.empty(4, device=dev, dtype=_f32)
            L.call("gens_blend_pack_t", L.ptr_table(w), nl, L.ptr(wstream, align=16), L.ptr(tab), L.ptr(sc), L.stream())
            L.call("gens_blend_views_t_dev", ctx.args[0], ctx.args[1], nl, ctx.args[3], ctx.args[4], ctx.args[5], ctx.args[6], views.nv, L.ptr(wstream),
                   L.ptr(tab), L.ptr(sc), L.ptr(pts), L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32), L.ptr(rgb), L.ptr(vis, torch.uint8), L.stream(),
                   nbytes=n * (24 + s), flops=n * flops, live=ctx.live, label="gens_blend_train_fwd")
        else:
            L.call("gens_blend_train_fwd", *ctx.args, L.ptr(rgb), L.ptr(vis, torch.uint8), L.stream(), nbytes=n * (24 + s), flops=n * flops, live=ctx.live)
        ctx.mark_non_differentiable(vis)
        return rgb, vis

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rgb, _g_vis):
        n, s, nl, f, flops, pshapes, fshapes, ishape = ctx.meta
        feats, imgs, w, views, pts, hw, idx, cnt = ctx.keep
        dev = pts.device
        rows = L.load().gens_blend_train_rows(n, views.nv)
        ins = [4, 16, 3 * f, 64, 32, 32, 32, 32, 37, 16, 8]
        outs = [16, f, 64, 32, 32, 33, 32, 1, 16, 8, 1]
        e = lambda *shape: torch.empty(*shape, device=dev, dtype=_f32)  # noqa: E731
        ev = lambda x: (x + 1) // 2 * 2  # noqa: E731      (even widths: the batched product then reads 8 bytes per lane)
        r_ops = [e(rows, ev(k + 1)) for k in ins]
        l_ops = [e(rows, ev(m)) for m in outs]
        want_maps = any(ctx.needs_input_grad[3 + 23:])
        g_feat = e(n, s, f) if want_maps else None
        s_part = e(rows // 32)
        L.call("gens_blend_train_bwd", *ctx.args, L.ptr(_c(g_rgb.to(_f32))), L.ptr_table(r_ops), L.ptr_table(l_ops), L.ptr(g_feat), L.ptr(s_part),
               L.stream(), nbytes=4 * rows * (sum(ins) + 11 + sum(outs)), flops=3 * n * flops, live=ctx.live)
        # [dW_l | db_l] = l_ops[l]^T r_ops[l]: eleven products over the same rows in one launch
        ms, ns = [ev(m) for m in outs], [ev(k + 1) for k in ins]
        mi, ni = L.int_table(ms), L.int_table(ns)
        ws = e(L.load().gens_gemm_tn_batch_workspace(11, mi, ni, rows))
        cc = e(sum(m * k for m, k in zip(ms, ns)))
        if cnt is None:
            L.call("gens_gemm_tn_batch", 11, L.ptr_table(l_ops), mi, L.ptr_table(r_ops), ni, mi, ni, rows, L.ptr(ws), L.ptr(cc), L.stream(),
                   nbytes=4 * rows * (sum(ms) + sum(ns)), flops=2 * rows * sum(m * k for m, k in zip(ms, ns)))
        else:       # floor(32 / S) points -> 32 operand rows per workgroup of the backward launch
            L.call("gens_gemm_tn_batch_live", 11, L.ptr_table(l_ops), mi, L.ptr_table(r_ops), ni, mi, ni, rows, L.ptr(cnt, torch.int32), 32 // s, 32,
                   L.ptr(ws), L.ptr(cc), L.stream(), nbytes=4 * rows * (sum(ms) + sum(ns)), flops=2 * rows * sum(m * k for m, k in zip(ms, ns)),
                   live=ctx.live, label="gens_gemm_tn_batch")
        # the 23 parameter gradients out of the product blocks in one launch, as views of one flat buffer (contiguous each)
        sizes = [math.prod(sh) if len(sh) else 1 for sh in pshapes]
        flat = e(sum(sizes))
        grads, off = [], 0
        for sh, sz in zip(pshapes, sizes):
            grads.append(flat[off:off + sz].view(sh))
            off += sz
        L.call("gens_blend_train_wgrad", L.ptr(cc), L.ptr(s_part), s_part.numel(), L.ptr(w[22]), f, L.ptr_table([g_.reshape(-1) for g_ in grads]), L.stream())
        g_imgs, g_feats = None, [None] * nl
        if want_maps:
            want_img = ctx.needs_input_grad[3 + 23]
            want_feat = any(ctx.needs_input_grad[3 + 24:])
            g_feats_t = [torch.zeros(sh, device=dev, dtype=_f32) for sh in fshapes] if want_feat else None
            g_imgs = torch.zeros(ishape, device=dev, dtype=_f32) if want_img else None
            L.call("gens_lookup_feature_bwd_idx", L.int_table(hw), nl, L.ptr(views.w2c), L.ptr(views.intr), views.nv, L.ptr(pts), L.ptr(g_feat),
                   L.ptr(idx, torch.int64), n, L.ptr(cnt, torch.int32), L.ptr_table(g_feats_t), L.ptr(g_imgs), L.stream(), label="gens_lookup_feature_bwd")
            if want_feat:
                g_feats = g_feats_t
        return (None, None, None, *grads, g_imgs, *g_feats)


def blend_train(net, views, pts, sel=None):
    """Colour of every point blended from the source views, differentiable with respect to the network and the maps:
    -> (rgb (N,3), vis (N,S) bool).  pts (N,3) device float32 (no gradient flows to the points, as in the reference's call).
    sel (StepPoints): only the selected ray samples are evaluated (count on the device); rgb / vis are sel's dense arrays."""
    pts = _c(pts.detach().reshape(-1, 3).to(_f32))
    rgb, vis = _BlendTrain.apply(pts, views, sel, *blend_params(net), views.imgs_tex, *views.feat_tex)
    return rgb, (vis if sel is not None else vis.bool())        # (a selection's flags stay uint8: the compositing launch reads them as they are)


class StepPoints:
    """The masked evaluation set of ONE training render (implicit_surface.py:174-191,256-257,484-497) with nothing read back to the host:
    dense rows [ray samples | always-evaluated points | pseudo points] in one (N, 3) buffer, the selected rows as a device-side index list +
    counts (gens_compact_points: the reference's nonzero + first-ten rescue), and the dense outputs of the two networks, whose unselected
    rows the same launch fills with the reference's defaults (Q8).  The same launch computes max(z_vals) (:301) and
    inv_s = clip(exp(10 variance), 1e-6, 1e6) (:206) into `scalars`."""

    def __init__(self, pts_all, valid_all, n_ray, n_always, n_src, z=None, variance=None):
        dev = pts_all.device
        self.pts = pts_all
        self.n, self.n_ray, self.n_always = int(pts_all.shape[0]), int(n_ray), int(n_always)
        n = self.n
        self.idx = torch.empty(n, device=dev, dtype=torch.int64)
        self.counts = torch.empty(3, device=dev, dtype=torch.int32)
        self.y, self.g, self.s = (torch.empty(n, k, device=dev, dtype=_f32) for k in (1, 3, 3))
        self.rgb = torch.empty(self.n_ray, 3, device=dev, dtype=_f32)
        self.vis = torch.empty(self.n_ray, n_src, device=dev, dtype=torch.uint8)
        self.scalars = torch.empty(4, device=dev, dtype=_f32)
        zc = None if z is None else _c(z.detach())
        scratch = torch.empty(L.load().gens_compact_points_scratch(n), device=dev, dtype=torch.int32)
        L.call("gens_compact_points", L.ptr(_c(valid_all), torch.uint8), self.n_ray, self.n_always, n, L.ptr(self.idx, torch.int64),
               L.ptr(self.counts, torch.int32), L.ptr(self.y), L.ptr(self.g), L.ptr(self.s), L.ptr(self.rgb), L.ptr(self.vis, torch.uint8), n_src,
               L.ptr(zc), 0 if zc is None else zc.numel(), L.ptr(None if variance is None else _c(variance.detach().reshape(1))), L.ptr(self.scalars),
               L.ptr(scratch, torch.int32), L.stream(), nbytes=n * 9)
