"""Autograd-aware operators of the GenS hot path, each a thin shim over one C-ABI entry point of libgens_hip.so.

The shims allocate outputs with torch (device memory + current stream only) and wire first / second order
derivatives the way the reference's Function pair does (models/modules/grid_sample_cuda/cuda_gridsample.py:71-123):
twice differentiable, outputs of the second backward are constants.  Citations are relative to /root/reference.

The operators live in submodules by kernel family; this package re-exports every name (private helpers included), so `gens_amd.ops.X`
is what it was when ops was one module:
    ops.base      Shared pieces of the operator shims: kernel selection, layout helpers, the texel copies of maps, VolumeSet, SceneCams.
    ops.volume    K1: Volume.agg_mean_var (volume.py:13-63), forward and backward, one level or a scene's pyramid.
    ops.lookup    K2 / K2'' (lookup_volume with first and second derivatives), K3 (nearest masks, ray points), K4 (lookup_feature).
    ops.rays      K5-K7 (hierarchical sampling), K8 (compositing and the step-boundary kernels around it), K9 (patch reads, surface patch warp), K10 (TV).
    ops.geometry  K11 (lattice points) and K12 (iso-surface extraction on the device).
    ops.sdf       K6 (the fused SDF network in inference) and K17 (the SDF network of a training step).
    ops.gemm      K14: a^T b for tall operands, the weight-gradient product of the training step.
    ops.conv3d    K15 / K16: the 3 x 3 x 3 convolutions and the instance norm + ReLU of the cost-volume U-Net.
    ops.blend     K7 (fused source-view look-up + BlendingNetwork in inference) and K18 (the same for a training step).
    ops.conv2d    K21: the depth-wise 2-D convolutions of the MnasNet trunk.
"""
from .base import *  # noqa: F401,F403
from .volume import *  # noqa: F401,F403
from .lookup import *  # noqa: F401,F403
from .rays import *  # noqa: F401,F403
from .geometry import *  # noqa: F401,F403
from .sdf import *  # noqa: F401,F403
from .gemm import *  # noqa: F401,F403
from .conv3d import *  # noqa: F401,F403
from .blend import *  # noqa: F401,F403
from .conv2d import *  # noqa: F401,F403
