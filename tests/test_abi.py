"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol that
include/gens_hip.h declares; the ctypes table mirrors the header; the product package never touches oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gens_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gens_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def libpath():
    path = os.path.join(ROOT, "gens_amd", "csrc", "libgens_hip.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.dirname(path), "-j4"])
    return path


def test_header_declares_the_hot_path():
    names = declared_functions()
    for must in ["gens_volume_build_fwd", "gens_volume_build_bwd", "gens_lookup_volume_fwd", "gens_lookup_volume_bwd",
                 "gens_lookup_volume_bwd2", "gens_lookup_mask_nearest", "gens_lookup_feature_fwd", "gens_lookup_feature_bwd",
                 "gens_upsample", "gens_merge_samples", "gens_composite_fwd", "gens_composite_bwd", "gens_patch_sample_fwd",
                 "gens_patch_sample_bwd", "gens_tv_fwd", "gens_tv_bwd", "gens_lattice_points"]:
        assert must in names


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in gens_hip.h but not exported"
    lib.gens_abi_version.restype = ctypes.c_int
    assert lib.gens_abi_version() == 12


def test_ctypes_table_covers_header(libpath):
    from gens_amd import lib as L
    declared = set(declared_functions()) - {"gens_last_error", "gens_abi_version", "gens_tv_blocks", "gens_sdf_train_stash_bytes", "gens_gemm_tn_batch_workspace", "gens_blend_train_rows",
                                              "gens_volume_build_bwd_levels_scratch_bytes", "gens_sdf_grad_stash_bytes", "gens_sdf_grad_f16_stash_bytes", "gens_scene_cams_floats", "gens_compact_points_scratch",
                                              "gens_batchnorm2d_scratch_doubles", "gens_blend_train_acc_parts", "gens_blend_train_acc_floats", "gens_blend_train_t_parts", "gens_lookup_scatter_bricks_scratch_bytes"}
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    L.load()


def test_argument_errors_are_reported_without_a_gpu(libpath):
    """Argument validation happens before any launch, so it can be exercised on the CPU box."""
    from gens_amd import lib as L
    lib = L.load()
    rc = lib.gens_volume_build_fwd(None, None, None, 1.0, 3, 30, 40, 16, 1, None, None, None)
    assert rc == -1 and b"null" in lib.gens_last_error()
    hw, ok, bad = L.int_table([480, 640]), L.int_table([256]), L.int_table([250])
    assert 0 < lib.gens_volume_build_bwd_levels_scratch_bytes(hw, ok, 1, 5) < 2 ** 26 and lib.gens_volume_build_bwd_levels_scratch_bytes(hw, bad, 1, 5) == 0
    rc = lib.gens_volume_build_bwd_levels(None, hw, ok, 1, None, None, 5, None, None, None, None, None, 0, None)
    assert rc == -1 and b"null" in lib.gens_last_error()
    rc = lib.gens_merge_samples(None, None, None, None, None, None, 4, 120, 16, None, None, None, None)
    assert rc == -2
    # the split-half value + gradient kernel: three or five levels (whole chunks of eight 1 KB pieces); one stash slot per (CU, wave)
    assert lib.gens_sdf_grad_f16_pieces(3) == 1016 and lib.gens_sdf_grad_f16_pieces(5) == 1288 and lib.gens_sdf_grad_f16_pieces(4) == 0
    assert lib.gens_sdf_grad_f16_stash_bytes() == 2048 * (16384 + 32 * 256 + 256)
    with pytest.raises(RuntimeError):
        L.call("gens_tv_fwd", None, None, 0, 0, 0, None, None)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gens_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\.*oracle", src, flags=re.M), f"{f} imports the oracle"


def test_product_fails_loudly_without_device_tensors(libpath):
    import torch
    from gens_amd import ops
    with pytest.raises(RuntimeError):
        ops.lookup_mask(torch.zeros(4, 3), [torch.ones(1, 1, 4, 4, 4)])


def test_new_entry_points_validate_their_arguments(libpath):
    from gens_amd import lib as L
    lib = L.load()
    assert lib.gens_mc_classify(None, 1, 8, 8, 0.0, None, None, None, None, None, None) == -1      # lattice must be >= 2 per axis
    assert lib.gens_mc_emit(None, 8, 8, 8, 0.0, None, 3, None, None, None, None, None, None, None, None) == -1
    assert lib.gens_pack_mask_bits(None, 0, None, None) == -1


def test_conv_and_norm_entry_points_validate_their_arguments(libpath):
    """K15 / K16: shape, stride, pointer and size checks happen before any launch."""
    import ctypes as C
    from gens_amd import lib as L
    lib = L.load()
    dims = (C.c_int * 3)(8, 8, 8)
    assert lib.gens_conv3d_gather(None, None, None, 8, 8, dims, 1, None, None) == -1 and b"null" in lib.gens_last_error()
    assert lib.gens_conv3d_gather(None, None, None, 8, 8, dims, 3, None, None) == -1 and b"stride" in lib.gens_last_error()
    assert lib.gens_conv3d_gather(None, None, None, 0, 8, dims, 1, None, None) == -1
    huge = (C.c_int * 3)(1024, 1024, 1024)                                            # 4 GiB per channel: beyond the 32-bit buffer offsets
    assert lib.gens_conv3d_scatter2(None, None, 8, 8, huge, None, None) == -1 and b"2 GiB" in lib.gens_last_error()
    assert lib.gens_conv3d_wgrad(None, None, 8, 8, dims, 1, None, None) == -1
    assert lib.gens_conv3d_wgrad_parts(8, 8, dims) >= 4 and lib.gens_conv3d_wgrad_parts(0, 8, dims) == 0
    big = (C.c_int * 3)(256, 256, 256)
    parts = lib.gens_conv3d_wgrad_parts(8, 8, big)
    assert parts % 4 == 0 and 64 <= parts <= 4096                                     # one partial per wave of every voxel range
    assert lib.gens_instnorm_blocks(8, 256 ** 3) * 8 <= 8192 + 8 and lib.gens_instnorm_blocks(3, 10) == 1 and lib.gens_instnorm_blocks(0, 10) == 0
    assert lib.gens_instnorm_stats(None, 8, 64, None, None) == -1
    assert lib.gens_instnorm_relu_fwd(None, None, 8, 0, None, None) == -1
    assert lib.gens_instnorm_relu_bwd_stats(None, None, None, 8, 64, None, None) == -1
    assert lib.gens_instnorm_relu_bwd(None, None, None, None, 70000, 64, None, None) == -1
    assert lib.gens_tv_bwd_scaled(None, None, 4, 4, 4, 1.0, None, None, None) == -1


def test_limits_are_checked_where_the_model_is_built():
    """Channel / level limits of the kernels surface in GenS.__init__ with a message that names the limit, not deep inside a kernel call."""
    import pytest
    from gens_amd.config import Conf, gens_model_conf
    from gens_amd.models.gens import _check_limits
    _check_limits(gens_model_conf())                                                     # the shipped configuration passes
    bad = gens_model_conf()
    bad = Conf({**bad, "reg_network": {"d_voluem": [8] * 5, "d_out": [8] * 5, "d_base": 8}})
    with pytest.raises(ValueError, match="4 channels"):
        _check_limits(bad)
    nine = gens_model_conf(volume_dims=tuple([8] * 9))
    with pytest.raises(ValueError, match="GENS_MAX_LEVELS"):
        _check_limits(nine)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                                  # one to five levels: fused kernels, no warning
        for n in range(1, 6):
            _check_limits(gens_model_conf(volume_dims=tuple([8] * n)))
    with pytest.warns(RuntimeWarning, match="1 to 5"):
        _check_limits(gens_model_conf(volume_dims=tuple([8] * 6), n_feature_levels=5))


def test_offset_views_are_refused_by_float4_consumers():
    import torch
    from gens_amd import lib as L, ops
    t = torch.zeros(9)[1:]
    assert ops.aligned16(t).data_ptr() % 16 == 0 and torch.equal(ops.aligned16(t), t)
    assert ops.aligned16(torch.zeros(8)).data_ptr() % 16 == 0
    assert L.ptr(None, align=16) is None
