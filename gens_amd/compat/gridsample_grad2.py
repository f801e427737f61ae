"""The reference's native extension `gridsample_grad2` (JIT-built from gridsample_cuda.cpp / gridsample_cuda.cu by `cuda_gridsample.py:5`) as a plain
ctypes binding of libgens_hip.so -- nothing else of gens_amd is imported.  A maintainer who keeps the reference's OWN `cuda_gridsample.py` (its Function
pairs, its asserts) replaces the `cpp_extension.load(...)` line by `from gens_amd.compat import gridsample_grad2`:

    grad2_2d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode, align_corners) -> [grad_grad_output, grad_input, grad_grid]
    grad2_3d(...)                                                                                          (gridsample_cuda.cpp:26-56)

`padding_mode` arrives as the reference passes it: the index of ['zeros', 'border'] (cuda_gridsample.py:32,83; a bool in the C++ signature).  The tensors
are the reference's (contiguous NCHW / NCDHW input and gradients, grid (N, ..., 2 | 3)); outputs are freshly allocated like the extension's
(gridsample_cuda.cu:553-555, 620-622: zeros_like).  `grad2_grad_input` may be None (cuda_gridsample.py:113-114 substitutes zeros; the kernel skips the term)."""
import ctypes
import os

import torch

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("GENS_HIP_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc", "libgens_hip.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C gens_amd/csrc` (there is no fallback)")
        _LIB = ctypes.CDLL(path)
        _LIB.gens_last_error.restype = ctypes.c_char_p
        _LIB.gens_grid_sample_bwd2.restype = ctypes.c_int
    return _LIB


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _grad2(ndim, grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode, align_corners):
    for t in (grad2_grad_grid, grad_output, input, grid):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError("gridsample_grad2: float32 device tensors only")
    inp, grd, go, ggg = input.contiguous(), grid.contiguous(), grad_output.contiguous(), grad2_grad_grid.contiguous()
    ggi = None if grad2_grad_input is None else grad2_grad_input.contiguous()
    gg_out, g_in, g_grid = torch.empty_like(go), torch.zeros_like(inp), torch.empty_like(grd)
    n, c = inp.shape[0], inp.shape[1]
    sizes = (ctypes.c_int * ndim)(*inp.shape[2:])
    n_out = grd[0].numel() // ndim if n > 0 else 0
    lib = _lib()
    rc = lib.gens_grid_sample_bwd2(_ptr(ggi), _ptr(ggg), _ptr(go), _ptr(inp), _ptr(grd), ctypes.c_int(ndim), ctypes.c_int(n), ctypes.c_int(c), sizes,
                                   ctypes.c_int64(n_out), ctypes.c_int(int(padding_mode)), ctypes.c_int(int(bool(align_corners))), _ptr(gg_out), _ptr(g_in),
                                   _ptr(g_grid), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gens_grid_sample_bwd2 failed (rc={rc}): {lib.gens_last_error().decode()}")
    return [gg_out, g_in, g_grid]


def grad2_2d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode, align_corners):
    return _grad2(2, grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode, align_corners)


def grad2_3d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode, align_corners):
    return _grad2(3, grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode, align_corners)
