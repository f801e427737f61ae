#!/usr/bin/env python3
"""Which torch operators (by input shape) carry the element-wise / reduction time of the full training step: torch.profiler over 5 steps
of scripts/train_step_bench.py --full's step, grouped by (operator, input shapes)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "--full", "--steps", "3"]
import scripts.train_step_bench as tsb  # noqa: E402

_orig_print = print
steps = {}


def main():
    # run the bench's own main() under the profiler: its first steps warm up, the profiler sees all of them (3 + 2)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        tsb.main()
    rows = prof.key_averages(group_by_input_shape=True)
    want = ("aten::sum", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::mul", "aten::copy_", "aten::cat", "aten::zeros", "aten::mean")
    sel = [r for r in rows if r.key in want]
    sel.sort(key=lambda r: -r.device_time_total)
    for r in sel[:40]:
        print(f"{r.key:14s} n={r.count:5d} dev {r.device_time_total / 5e3:8.2f} ms/step  {str(r.input_shapes)[:110]}")


if __name__ == "__main__":
    main()
