"""ctypes binding of libgens_hip.so (C ABI declared in include/gens_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this module raises.  PyTorch is used by
the callers only to own device memory and the current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GENS_HIP_LIB", os.path.join(_HERE, "csrc", "libgens_hip.so"))   # env override: development builds

MAX_LEVELS = 8
MAX_VIEWS = 16
LAYOUT_PLANAR = 0
LAYOUT_PACKED = 1

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_f = C.c_float
_pp = C.POINTER(C.c_void_p)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)


class CompositeIn(C.Structure):
    _fields_ = [(n, _p) for n in ("rays_o", "rays_d", "z", "sdf", "grad", "color", "smooth", "voxel_mask", "src_vis", "inv_s", "z_max")] + [
        ("n_rays", _l), ("n", _i), ("n_src", _i), ("sample_dist", _f), ("cos_anneal", _f), ("rot", _f * 9), ("rot_dev", _p), ("cos_anneal_dev", _p)]


class LossArgs(C.Structure):
    _fields_ = [("color", _p), ("target", _p), ("valid", _p), ("b", _l), ("sparse", _p), ("n_sparse", _l), ("sparse_scale", _f), ("pseudo", _p),
                ("n_pseudo", _l), ("ncc", _p), ("mid_in", _p), ("depth", _p), ("pseudo_depth_t", _p), ("depth_t", _p), ("ge", _p), ("se", _p), ("tv", _p)] + [
        (n, _f) for n in ("w_color", "w_igr", "w_sparse", "w_mfc", "w_smooth", "w_tv", "w_pseudo_sdf", "w_pseudo_depth")] + [
        (n, _p) for n in ("out", "g", "g_color", "g_sparse", "g_pseudo", "g_ncc", "g_depth", "g_scalars")]


class CompositeOut(C.Structure):
    _fields_ = [(n, _p) for n in ("color", "normal", "depth", "wsum", "wmax", "mid_in", "sdf_depth", "z_cross", "eik_num", "eik_den",
                                  "smooth_vec", "valid", "cross_idx", "weights", "inside", "pts_cross")]


class CompositeGrad(C.Structure):
    _fields_ = [(n, _p) for n in ("g_color", "g_normal", "g_depth", "g_weights", "g_wsum", "g_eik_num", "g_smooth_vec", "g_z_cross",
                                  "weights", "cross_idx", "smooth_vec", "g_sdf", "g_grad", "g_col", "g_smooth", "g_inv_s", "g_gradient_error",
                                  "g_smooth_error", "finish")]


# name -> argtypes, mirroring include/gens_hip.h declaration by declaration
SIGNATURES = {
    "gens_pack_nchw": [_p, _p, _i, _i, _i, _i, _p],
    "gens_unpack_nhwc": [_p, _p, _i, _i, _i, _i, _p],
    "gens_pack_volume": [_p, _p, _i, _i, _i, _p],
    "gens_volume_build_fwd": [_p, _p, _p, _f, _i, _i, _i, _i, _i, _p, _p, _p],
    "gens_volume_build_bwd": [_p, _p, _p, _f, _i, _i, _i, _i, _p, _p, _p],
    "gens_selftest_division": [_p, _p],
    "gens_volume_build_levels": [_pp, _ip, _ip, _i, _p, _pp, _i, _i, _pp, _pp, _pp, _p],
    "gens_volume_build_levels_bits": [_pp, _ip, _ip, _i, _p, _pp, _i, _i, _pp, _pp, _pp, _pp, _p],
    "gens_volume_build_bwd_levels": [_pp, _ip, _ip, _i, _p, _pp, _i, _pp, _pp, _pp, _pp, _p, _l, _p],
    "gens_gemm_tn_slabs": [_l, _i, _i],
    "gens_gemm_tn": [_p, _p, _l, _i, _i, _p, _p, _p],
    "gens_conv3d_gather": [_p, _p, _p, _i, _i, _ip, _i, _p, _p],
    "gens_conv3d_scatter2": [_p, _p, _i, _i, _ip, _p, _p],
    "gens_conv3d_wgrad_parts": [_i, _i, _ip],
    "gens_conv3d_wgrad_parts_strided": [_i, _i, _ip, _i],
    "gens_conv3d_wgrad": [_p, _p, _i, _i, _ip, _i, _p, _p],
    "gens_instnorm_blocks": [_i, _l],
    "gens_instnorm_stats": [_p, _i, _l, _p, _p],
    "gens_instnorm_finish": [_p, _i, _l, C.c_double, _i, _p, _p],
    "gens_instnorm_relu_fwd": [_p, _p, _i, _l, _p, _p],
    "gens_instnorm_relu_add_fwd": [_p, _p, _p, _i, _l, _p, _p],
    "gens_instnorm_relu_bwd_stats": [_p, _p, _p, _i, _l, _p, _p],
    "gens_instnorm_relu_bwd": [_p, _p, _p, _p, _i, _l, _p, _p],
    "gens_lookup_volume_fwd": [_pp, _ip, _i, _i, _p, _l, _p, _p],
    "gens_lookup_volume_bwd": [_pp, _ip, _i, _i, _p, _p, _l, _pp, _p, _p],
    "gens_lookup_volume_bwd_bricks": [_pp, _ip, _i, _i, _p, _p, _l, _pp, _p, _p, _l, _p],
    "gens_lookup_volume_bwd2": [_pp, _ip, _i, _i, _p, _p, _p, _pp, _l, _p, _pp, _p, _p],
    "gens_lookup_volume_bwd2_bricks": [_pp, _ip, _i, _i, _p, _p, _p, _pp, _l, _p, _pp, _p, _p, _l, _p],
    "gens_lookup_mask_nearest": [_pp, _ip, _i, _p, _l, _p, _p, _p],
    "gens_ray_points": [_p, _p, _p, _l, _i, _i, _f, _pp, _ip, _i, _i, _p, _p, _p],
    "gens_pack_mask_bits": [_p, _l, _p, _p],
    "gens_lookup_feature_fwd": [_pp, _ip, _i, _p, _p, _p, _p, _i, _p, _l, _p, _p, _p, _p],
    "gens_lookup_feature_bwd": [_ip, _i, _p, _p, _i, _p, _p, _l, _pp, _p, _p],
    "gens_upsample": [_p, _p, _p, _p, _l, _i, _i, _f, _pp, _ip, _i, _i, _p, _p, _p, _p, _p],
    "gens_merge_samples": [_p, _p, _p, _p, _p, _p, _l, _i, _i, _p, _p, _p, _p],
    "gens_merge_upsample": [_p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _f, _f, _pp, _ip, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "gens_composite_fwd": [C.POINTER(CompositeIn), C.POINTER(CompositeOut), _p],
    "gens_composite_bwd": [C.POINTER(CompositeIn), C.POINTER(CompositeGrad), _p],
    "gens_patch_sample_fwd": [_p, _i, _i, _i, _p, _l, _p, _p],
    "gens_patch_sample_bwd": [_p, _i, _i, _i, _p, _p, _l, _p, _p],
    "gens_upsample2d_into": [_p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _p],
    "gens_upsample2d_cat": [_pp, _ip, _i, _i, _p, _i, _i, _i, _p],
    "gens_select_views": [_pp, _pp, _ip, _ip, _i, _p, _i, _p],
    "gens_tv_fwd": [_p, _p, _i, _i, _i, _p, _p],
    "gens_tv_bwd": [_p, _p, _i, _i, _i, _f, _p, _p],
    "gens_tv_bwd_scaled": [_p, _p, _i, _i, _i, _f, _p, _p, _p],
    "gens_lattice_points": [_fp, _fp, _i, _l, _l, _p, _p],
    "gens_blend_views": [_pp, _ip, _i, _p, _p, _p, _p, _i, _pp, _fp, _p, _p, _l, _p, _p, _p, _p],
    "gens_blend_views4": [_pp, _ip, _i, _p, _p, _p, _p, _i, _p, _p, _fp, _p, _p, _l, _p, _p, _p, _p],
    "gens_blend_views4_groups": [_i],
    "gens_blend_views_t": [_pp, _ip, _i, _p, _p, _p, _p, _i, _p, _p, _fp, _p, _p, _l, _p, _p, _p, _p],
    "gens_blend_views_t_groups": [_i],
    "gens_blend_pack_t": [_pp, _i, _p, _p, _p, _p],
    "gens_blend_views_t_dev": [_pp, _ip, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p],
    "gens_compact_valid": [_p, _l, _p, _p, _p, _p],
    "gens_sdf_value": [_pp, _ip, _i, _p, _p, _f, _f, _p, _p, _l, _p, _p, _p],
    "gens_sdf_value_groups": [_i],
    "gens_sdf_grad": [_pp, _ip, _i, _p, _p, _f, _f, _p, _p, _l, _p, _p, _p, _p, _p],
    "gens_sdf_grad_groups": [_i],
    "gens_sdf_grad_f16": [_pp, _ip, _i, _p, _p, _f, _f, _f, _p, _p, _l, _p, _p, _p, _p, _p, _p],
    "gens_sdf_grad_f16_pieces": [_i],
    "gens_sdf_value_f16": [_pp, _ip, _i, _p, _p, _f, _f, _p, _p, _l, _p, _p, _p, _p],
    "gens_sdf_value_f16_units": [_i],
    "gens_lncc_fwd": [_p, _p, _l, _i, _i, _i, _p, _p, _p],
    "gens_lncc_bwd": [_p, _p, _p, _p, _l, _i, _i, _i, _p, _p, _p],
    "gens_mc_classify": [_p, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p],
    "gens_mc_emit": [_p, _i, _i, _i, _f, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "gens_sdf_mlp": [_pp, _ip, _i, _pp, _pp, _p, _p, _f, _f, _p, _p, _l, _p, _p, _p, _p],
    "gens_sdf_mlp_dev": [_pp, _ip, _i, _pp, _pp, _p, _p, _f, _p, _p, _l, _p, _p, _p, _p],
    "gens_blend_train_fwd": [_pp, _ip, _i, _p, _p, _p, _p, _i, _pp, _p, _p, _l, _p, _p, _p, _p],
    "gens_blend_train_bwd": [_pp, _ip, _i, _p, _p, _p, _p, _i, _pp, _p, _p, _l, _p, _p, _pp, _pp, _p, _p, _p],
    "gens_blend_train_bwd_acc": [_pp, _ip, _i, _p, _p, _p, _p, _i, _pp, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p],
    "gens_blend_train_bwd_t": [_pp, _ip, _i, _p, _p, _p, _p, _i, _pp, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p],
    "gens_blend_train_bwd_t_dump": [_pp, _ip, _i, _p, _p, _p, _p, _i, _pp, _p, _p, _l, _p, _p, _p, _p, _p, _p, _pp, _pp, _p],
    "gens_lookup_feature_bwd_idx": [_ip, _i, _p, _p, _i, _p, _p, _p, _l, _p, _pp, _p, _p],
    "gens_gemm_tn_batch_live": [_i, _pp, _ip, _pp, _ip, _ip, _ip, _l, _p, _i, _i, _p, _p, _p],
    "gens_sdf_train_pack": [_pp, _pp, _i, _pp, _pp, _p],
    "gens_sdf_train_pack_wn": [_pp, _pp, _pp, _i, _pp, _pp, _pp, _p, _p, _p],
    "gens_sdf_train_wgrad": [_pp, _pp, _i, _p, _p, _pp, _pp, _pp, _p],
    "gens_sdf_train_fwd": [_pp, _ip, _i, _pp, _pp, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p],
    "gens_sdf_train_bwd": [_pp, _ip, _i, _pp, _pp, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "gens_gemm_tn_batch": [_i, _pp, _ip, _pp, _ip, _ip, _ip, _l, _p, _p, _p],
    "gens_sdf_train_scatter": [_ip, _i, _p, _p, _p, _p, _p, _p, _p, _l, _p, _pp, _p],
    "gens_scene_setup": [_p, _p, _i, _p, _p],
    "gens_coarse_z": [_p, _p, _i, _p, _p, _l, _i, _p, _p],
    "gens_sdf_grad_stash_reset": [_p, _p],
    "gens_depthwise_conv2d_fwd": [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "gens_depthwise_conv2d_dgrad": [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "gens_depthwise_conv2d_wgrad_parts": [_i, _i, _i, _i, _i, _i],
    "gens_depthwise_conv2d_wgrad": [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "gens_batchnorm2d_train_fwd": [_p, _p, _p, _i, _i, _i, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p],
    "gens_batchnorm2d_train_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p],
    "gens_grid_sample_fwd": [_p, _p, _i, _i, _i, _ip, _l, _i, _i, _p, _p],
    "gens_grid_sample_bwd": [_p, _p, _p, _i, _i, _i, _ip, _l, _i, _i, _p, _p, _p],
    "gens_grid_sample_bwd2": [_p, _p, _p, _p, _p, _i, _i, _i, _ip, _l, _i, _i, _p, _p, _p, _p],
    "gens_sdf_grad_f16_stash_reset": [_p, _p],
    "gens_blend_train_wgrad": [_p, _p, _i, _p, _i, _pp, _p],
    "gens_loss_fwd": [_p, _p],
    "gens_loss_bwd": [_p, _p],
    "gens_composite_finish_fwd": [_p, _p, _p, _l, _p, _p],
    "gens_composite_finish_bwd": [_p, _l, _p, _p, _p, _p, _p, _l, _l, _p],
    "gens_patch_warp_fwd": [_p, _p, _p, _p, _l, _p, _p, _p, _i, _p, _i, _i, _i, _i, _p, _p, _p],
    "gens_patch_warp_bwd": [_p, _p, _p, _p, _l, _p, _p, _p, _i, _p, _i, _i, _i, _i, _p, _p, _p],
    "gens_pack_maps": [_pp, _pp, _ip, _i, _p],
    "gens_unpack_maps": [_pp, _pp, _ip, _i, _p],
    "gens_compact_points": [_p, _l, _l, _l, _p, _p, _p, _p, _p, _p, _p, _i, _p, _l, _p, _p, _p, _p],
    "gens_tv_levels_blocks": [_ip, _i],
    "gens_tv_levels_fwd": [_pp, _pp, _ip, _i, _p, _p, _p],
    "gens_tv_levels_bwd": [_pp, _pp, _ip, _i, _p, _p, _pp, _p],
}

_lib = None


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C gens_amd/csrc` (or __graft_entry__.build()); "
                           "gens_amd has no CPU / PyTorch fallback for the hot path")
    lib = C.CDLL(LIB_PATH)
    lib.gens_last_error.restype = C.c_char_p
    lib.gens_last_error.argtypes = []
    lib.gens_abi_version.restype = _i
    lib.gens_abi_version.argtypes = []
    lib.gens_tv_blocks.restype = _i
    lib.gens_tv_blocks.argtypes = [_l]
    lib.gens_volume_build_bwd_levels_scratch_bytes.restype = _l
    lib.gens_volume_build_bwd_levels_scratch_bytes.argtypes = [_ip, _ip, _i, _i]
    lib.gens_sdf_train_stash_bytes.restype = _l
    lib.gens_sdf_grad_stash_bytes.restype = _l
    lib.gens_sdf_grad_stash_bytes.argtypes = []
    lib.gens_sdf_grad_f16_stash_bytes.restype = _l
    lib.gens_sdf_grad_f16_stash_bytes.argtypes = []
    lib.gens_sdf_train_stash_bytes.argtypes = [_l, _i]
    lib.gens_blend_train_rows.restype = _l
    lib.gens_blend_train_rows.argtypes = [_l, _i]
    lib.gens_blend_train_acc_parts.restype = _i
    lib.gens_blend_train_acc_parts.argtypes = [_l, _i]
    lib.gens_blend_train_acc_floats.restype = _i
    lib.gens_blend_train_acc_floats.argtypes = [_i]
    lib.gens_blend_train_t_parts.restype = _i
    lib.gens_blend_train_t_parts.argtypes = [_l, _i]
    lib.gens_lookup_scatter_bricks_scratch_bytes.restype = _l
    lib.gens_lookup_scatter_bricks_scratch_bytes.argtypes = [_l]
    lib.gens_gemm_tn_batch_workspace.restype = _l
    lib.gens_gemm_tn_batch_workspace.argtypes = [_i, _ip, _ip, _l]
    lib.gens_scene_cams_floats.restype = _l
    lib.gens_scene_cams_floats.argtypes = [_i]
    lib.gens_batchnorm2d_scratch_doubles.restype = _l
    lib.gens_batchnorm2d_scratch_doubles.argtypes = [_i, _i, _i]
    lib.gens_compact_points_scratch.restype = _l
    lib.gens_compact_points_scratch.argtypes = [_l]
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = _i
        fn.argtypes = args
    _lib = lib
    return lib


_profile = None   # None, or a list of (name, start_event, end_event, algorithmic_bytes)


_profile_only = None


def profile_begin(only=None):
    """Start recording a HIP event pair around every kernel launch (on the stream the kernel is enqueued on).
    only: optional set of entry-point names to restrict the recording to (an event pair costs ~15 us of host time)."""
    global _profile, _profile_only
    _profile = []
    _profile_only = set(only) if only else None


def profile_end(raw=False, per_launch=False):
    """Stop recording; -> {kernel: {"launches", "ms", "bytes"}} (synchronises).  raw=True: the per-launch list
    [(kernel, ms, algorithmic bytes, flops)] in launch order instead.  per_launch=True: (the table, {kernel: [ms of every launch]})."""
    global _profile
    rec, _profile = _profile or [], None
    torch.cuda.synchronize()
    if raw:
        return [(name, s.elapsed_time(e), nbytes, flops) for name, s, e, nbytes, flops, _live in rec]
    out, each = {}, {}
    for name, s, e, nbytes, flops, live in rec:
        d = out.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0})
        frac = 1.0
        if live is not None:        # (device count tensor, launched points): algorithmic work is that of the points really evaluated
            frac = min(1.0, float(live[0].item()) / max(1, live[1]))
        ms = s.elapsed_time(e)
        d["launches"] += 1
        d["ms"] += ms
        d["bytes"] += int(nbytes * frac)
        d["flops"] += int(flops * frac)
        each.setdefault(name, []).append(ms)
    return (out, each) if per_launch else out


def call(name, *args, nbytes=0, flops=0, live=None, label=None):
    """Invoke one C-ABI entry point; `nbytes` / `flops` = algorithmic HBM bytes / FLOPs of this launch.
    label: key of this launch in the profile table when one entry point stands for several device kernels (default: name)."""
    lib = load()
    key = label or name
    if _profile is not None and (_profile_only is None or key in _profile_only):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(lib, name)(*args)
        e.record()
        _profile.append((key, s, e, int(nbytes), int(flops), (live[0].clone(), live[1]) if live is not None else None))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.gens_last_error().decode()}")


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32, align=None):
    """Device pointer of a contiguous CUDA(HIP) tensor (None -> NULL).  align=16 for buffers a kernel reads / writes as float4."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("gens_amd kernels need device tensors (no CPU path)")
    if t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")
    if align and t.data_ptr() % align != 0 and t.numel() > 0:
        # texels, packed volumes and the K15 / K16 tensors are accessed as float4 / float2: a view that starts in the middle of an allocation
        # (flat[1:].view(...)) would fault or tear; torch's own allocations are 256-byte aligned.  ops.aligned16() clones such a view.
        raise RuntimeError(f"tensor storage is not {align}-byte aligned (offset view, data_ptr % {align} = {t.data_ptr() % align}): pass a .clone()")
    return C.c_void_p(t.data_ptr())


def ptr_table(tensors, dtype=torch.float32, align=None):
    """HOST array of device pointers (None -> NULL table)."""
    if tensors is None:
        return None
    arr = (C.c_void_p * len(tensors))()
    for k, t in enumerate(tensors):
        arr[k] = None if t is None else ptr(t, dtype, align).value
    return C.cast(arr, _pp)


def int_table(values):
    flat = [int(v) for v in values]
    return (C.c_int * len(flat))(*flat)
