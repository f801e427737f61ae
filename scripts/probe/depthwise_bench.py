"""K21 at the MnasNet trunk's shapes (5 views, 480 x 640 input): time and achieved HBM rate of forward / data gradient / weight gradient per layer,
next to MIOpen (torch.nn.functional.conv2d) on the same tensors."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import ops  # noqa: E402

LAYERS = [(32, 240, 320, 3, 1), (48, 240, 320, 3, 2), (72, 120, 160, 3, 1), (72, 120, 160, 5, 2), (120, 60, 80, 5, 1), (240, 60, 80, 5, 2),
          (480, 30, 40, 5, 1), (480, 30, 40, 3, 1), (576, 30, 40, 3, 1), (576, 30, 40, 5, 2), (1152, 15, 20, 5, 1), (1152, 15, 20, 3, 1)]


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "miopen_fwd": 0.0, "miopen_bwd": 0.0}
for c, h, w, k, s in LAYERS:
    x = torch.randn(5, c, h, w, device="cuda", requires_grad=True)
    wt = torch.randn(c, 1, k, k, device="cuda", requires_grad=True)
    out = ops.depthwise_conv2d(x, wt, s)
    go = torch.randn_like(out)
    t_f = timed(lambda: ops.depthwise_conv2d(x, wt, s))
    t_d = timed(lambda: torch.autograd.grad(ops.depthwise_conv2d(x, wt.detach(), s), x, go)) - t_f
    t_w = timed(lambda: torch.autograd.grad(ops.depthwise_conv2d(x.detach(), wt, s), wt, go)) - t_f
    m_f = timed(lambda: F.conv2d(x, wt, None, s, k // 2, 1, c))
    m_b = timed(lambda: torch.autograd.grad(F.conv2d(x, wt, None, s, k // 2, 1, c), [x, wt], go)) - m_f
    nb = 4 * (x.numel() + out.numel())
    print(f"c={c:5d} {h:3d}x{w:<3d} k{k} s{s}: fwd {t_f:6.1f} us ({nb / t_f / 1e3:6.0f} GB/s)  dgrad {t_d:6.1f}  wgrad {t_w:6.1f}   MIOpen fwd {m_f:7.1f} bwd {m_b:7.1f} us")
    for key, v in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w), ("miopen_fwd", m_f), ("miopen_bwd", m_b)):
        tot[key] += v
print("sum over the listed layers (us):", {k: round(v, 1) for k, v in tot.items()})
