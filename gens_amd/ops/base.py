"""Autograd-aware operators of the GenS hot path, each a thin shim over one C-ABI entry point of libgens_hip.so.

The shims allocate outputs with torch (device memory + current stream only) and wire first / second order
derivatives the way the reference's Function pair does (models/modules/grid_sample_cuda/cuda_gridsample.py:71-123):
twice differentiable, outputs of the second backward are constants.  Citations are relative to /root/reference.
"""
import ctypes as C
import os
import math

import torch

from .. import lib as L

_f32 = torch.float32


class KernelChoice:
    """Which generation of a fused kernel an operator launches -- plain attributes, set once from the environment at import (the switches
    of INTEGRATION.md) and changed by assignment afterwards (tests: monkeypatch.setattr(ops.kernels, ...)).  The operators read these
    attributes; nothing on a launch path reads os.environ.
        sdf_value / sdf_grad   "transposed" (k6t / k6g: register-chained, the default) | "rowmajor" (k6_sdfmlp.hip: cross-check, other shapes)
        blend                  "transposed" (k7t, two to four source views) | "rowmajor" (k7_blend.hip)
        blend_train_fwd        "transposed" (the training step's forward through k7t + gens_blend_pack_t) | "rowmajor" (k18's own forward)
        blend_train_wgrad      "inside" (the weight-gradient sums inside the backward launch, gens_blend_train_bwd_acc) | "rows" (operand rows + K14)
        blend_train_bwd        "transposed" (two to four source views: gens_blend_train_bwd_t, a wave per 16 rows, weights in LDS) | "rowmajor" (k18's
                               32-row workgroups; always for other view counts)
        sdf_grad_f16           True: under sdf_precision "f16x2" the value + gradient pass runs on the split-half kernel too (k6gh) | False: float32
        k1_bwd                 "auto" (all levels on the image-tile kernel) | "window" (the wave-window kernel, level by level)
        tex_cache              texel copies kept on the map tensors (pack_maps)
        select_views           fine-tuning takes a step's views out of the frozen maps and their layouts in one launch (gens_select_views) | torch.index_select
        k2_bricks_min          points from which the stand-alone look-up's volume-gradient scatter runs brick by brick (gens_lookup_volume_bwd_bricks)"""

    def __init__(self, env=os.environ):
        self.sdf_value = "rowmajor" if env.get("GENS_SDF_VALUE_ROWMAJOR") else "transposed"
        self.sdf_grad = "rowmajor" if env.get("GENS_SDF_GRAD_ROWMAJOR") else "transposed"
        self.sdf_grad_f16 = not env.get("GENS_SDF_GRAD_F32_ONLY")
        self.blend = "rowmajor" if env.get("GENS_BLEND_ROWMAJOR") else "transposed"
        self.blend_train_fwd = "rowmajor" if env.get("GENS_BLEND_TRAIN_ROWMAJOR") else "transposed"
        self.blend_train_wgrad = "rows" if env.get("GENS_K18_OPERAND_ROWS") else "inside"
        self.blend_train_bwd = "rowmajor" if env.get("GENS_BLEND_TRAIN_BWD_ROWMAJOR") or env.get("GENS_K18_OPERAND_ROWS") else "transposed"
        self.k1_bwd = "window" if env.get("GENS_K1_BWD_WINDOW") else "auto"
        self.tex_cache = not env.get("GENS_NO_TEX_CACHE")
        self.select_views = not env.get("GENS_NO_SELECT_VIEWS")
        self.k2_bricks_min = int(env.get("GENS_K2_BRICKS_MIN", "196608"))


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


def reset_sdf_grad_stashes():
    """Re-zero every stash of this process on the current stream (gens_sdf_grad_stash_reset / gens_sdf_grad_f16_stash_reset): the lock words of
    slots a failed launch may have left taken.  A wave that finds its slot taken waits a bounded time and traps -- a loud failure, not a hang."""
    for key, buf in _SDF_GRAD_STASH.items():
        with torch.cuda.device(key):
            L.call("gens_sdf_grad_stash_reset", L.ptr(buf, torch.uint8), L.stream())
    for key, buf in _SDF_GRAD_F16_STASH.items():
        with torch.cuda.device(key):
            L.call("gens_sdf_grad_f16_stash_reset", L.ptr(buf, torch.uint8), L.stream())


_SDF_GRAD_F16_STASH = {}


def sdf_grad_f16_stash(device):
    """gens_sdf_grad_f16's (CU, wave)-private slots (softplus' of one layer and the trilinear Jacobians): one zeroed buffer per device."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    buf = _SDF_GRAD_F16_STASH.get(key)
    if buf is None:
        buf = _SDF_GRAD_F16_STASH[key] = torch.zeros(L.load().gens_sdf_grad_f16_stash_bytes(), device=torch.device("cuda", key), dtype=torch.uint8)
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


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]      # private helpers travel too: the package namespace is the old module's
