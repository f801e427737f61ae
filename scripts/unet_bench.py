#!/usr/bin/env python3
"""The cost-volume U-Net (RegNetwork: K15 convolutions + K16 instance-norm/ReLU) at the benchmark pyramid, 8-channel cost volumes
256^3 / 128^3 / 64^3: forward + backward time, and every K15 / K16 launch timed with HIP events on its stream
(algorithmic FLOP/s against the 157.3 TFLOP/s float32 vector peak for the convolutions, algorithmic bytes/s against 8 TB/s for K16).

    python scripts/unet_bench.py [--out profiles/rNN_unet_kernels.json]
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gens_amd import lib as L  # noqa: E402
from gens_amd.config import gens_model_conf  # noqa: E402
from gens_amd.models.modules.reg_network import RegNetwork  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dims = (256, 128, 64)
    torch.manual_seed(0)
    net = RegNetwork(gens_model_conf(volume_dims=dims)["reg_network"]).to(dev).train()
    vols = [torch.randn(1, 8, d, d, d, device=dev, requires_grad=True) for d in dims]
    cots = [torch.randn(1, 4, d, d, d, device=dev) for d in dims]

    def step():
        outs = net(vols)
        torch.autograd.backward(outs, cots)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.iters):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            net(vols)
        torch.cuda.synchronize()
        fwd = (time.perf_counter() - t0) / args.iters * 1e3
    L.profile_begin()
    for _ in range(args.iters):
        step()
    rec = L.profile_end(raw=True)
    per = {}
    for name, ms, nbytes, flops in rec:
        d = per.setdefault((name, nbytes, flops), [])
        d.append(ms)
    rows = []
    for (name, nbytes, flops), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        us = statistics.median(v) * 1e3
        rows.append({"kernel": name, "launches_per_step": len(v) // args.iters, "median_us": round(us, 1), "algorithmic_MB": round(nbytes / 1e6, 1),
                     "GFLOP": round(flops / 1e9, 2), "TFLOP_s": round(flops / us / 1e6, 1), "GB_s": round(nbytes / us / 1e3, 0),
                     "frac_fp32_vector_peak": round(flops / us / 1e6 / 157.3, 3), "frac_hbm_peak": round(nbytes / us / 1e3 / 8000, 3),
                     "step_ms": round(sum(v) / args.iters, 3)})
    out = {"workload": "RegNetwork d_base 8, cost volumes 8 x {256,128,64}^3, forward + backward", "fwd_ms": round(fwd, 2),
           "fwd_bwd_ms_median": round(statistics.median(ts), 2), "in_k15_k16_ms": round(sum(r["step_ms"] for r in rows), 2), "launch_shapes": rows}
    print(json.dumps({k: v for k, v in out.items() if k != "launch_shapes"}))
    for r in rows[:14]:
        print(f'{r["kernel"]:30s} x{r["launches_per_step"]:2d} {r["median_us"]:9.1f} us  {r["GFLOP"]:7.2f} GFLOP {r["TFLOP_s"]:6.1f} TFLOP/s  '
              f'{r["algorithmic_MB"]:7.1f} MB {r["GB_s"]:7.0f} GB/s  step {r["step_ms"]:.2f} ms')
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
