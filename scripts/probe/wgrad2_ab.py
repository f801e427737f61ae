"""conv3d weight gradient, STRIDE 2: the matrix-core kernel against the vector-ALU kernel (GENS_K15_NO_MFMA_WGRAD2) at the U-Net's layer shapes
(P = the coarse tensor, Q = the fine one: Conv3d(cq -> cp, stride 2) and ConvTranspose3d(cp -> cq, stride 2) share the relation)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L
from gens_amd.ops.conv3d import _conv_wgrad  # noqa: E402

L.load()
for cp, cq, d in ((16, 8, 128), (32, 16, 64), (16, 8, 64), (5, 12, 64)):
    g = torch.Generator(device="cuda").manual_seed(1)
    p = torch.randn(cp, d, d, d, device="cuda", generator=g)
    q = torch.randn(cq, 2 * d, 2 * d, 2 * d, device="cuda", generator=g)
    res = {}
    for name in ("mfma", "valu"):
        if name == "valu":
            os.environ["GENS_K15_NO_MFMA_WGRAD2"] = "1"
        else:
            os.environ.pop("GENS_K15_NO_MFMA_WGRAD2", None)
        out = _conv_wgrad(p, q, 2)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            _conv_wgrad(p, q, 2)
        e.record()
        torch.cuda.synchronize()
        res[name] = (s.elapsed_time(e) / 10, out)
    os.environ.pop("GENS_K15_NO_MFMA_WGRAD2", None)
    fl = 2 * 27 * cp * cq * d ** 3
    err = float((res["mfma"][1] - res["valu"][1]).abs().max() / res["valu"][1].abs().max())
    print(f"{cp:2d} x {cq:2d} @ coarse {d}^3: mfma {res['mfma'][0]:7.3f} ms ({fl / res['mfma'][0] / 1e9:6.1f} TFLOP/s)   valu {res['valu'][0]:7.3f} ms "
          f"({fl / res['valu'][0] / 1e9:6.1f})   rel diff {err:.1e}")
