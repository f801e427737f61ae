"""conv3d weight gradient, stride 1: the matrix-core kernel against the vector-ALU kernel (GENS_K15_NO_MFMA_WGRAD) at the U-Net's layer shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L
from gens_amd.ops.conv3d import _conv_wgrad  # noqa: E402

L.load()
for cp, cq, d in ((8, 8, 256), (4, 8, 256), (8, 8, 128), (4, 8, 128), (16, 16, 64), (4, 16, 64)):
    g = torch.Generator(device="cuda").manual_seed(1)
    p = torch.randn(cp, d, d, d, device="cuda", generator=g)
    q = torch.randn(cq, d, d, d, device="cuda", generator=g)
    res = {}
    for name in ("mfma", "valu"):
        if name == "valu":
            os.environ["GENS_K15_NO_MFMA_WGRAD"] = "1"
        else:
            os.environ.pop("GENS_K15_NO_MFMA_WGRAD", None)
        out = _conv_wgrad(p, q, 1)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            _conv_wgrad(p, q, 1)
        e.record()
        torch.cuda.synchronize()
        res[name] = (s.elapsed_time(e) / 10, out)
    os.environ.pop("GENS_K15_NO_MFMA_WGRAD", None)
    fl = 2 * 27 * cp * cq * d ** 3
    err = float((res["mfma"][1] - res["valu"][1]).abs().max() / res["valu"][1].abs().max())
    print(f"{cp:2d} x {cq:2d} @ {d}^3: mfma {res['mfma'][0]:7.3f} ms ({fl / res['mfma'][0] / 1e9:6.1f} TFLOP/s)   valu {res['valu'][0]:7.3f} ms ({fl / res['valu'][0] / 1e9:6.1f})   rel diff {err:.1e}")
