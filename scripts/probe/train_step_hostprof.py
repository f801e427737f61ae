#!/usr/bin/env python3
"""Host-side view of one training step (torch.profiler): which operators cost host time and launches.
    python scripts/probe/train_step_hostprof.py [--finetune]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scripts import train_step_bench as B  # noqa: E402


def main():
    from torch.profiler import ProfilerActivity, profile, record_function
    prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False)
    argv = [a for a in sys.argv[1:]]
    # run the bench once normally for warm-up, then a few profiled steps by monkey-patching its timing loop
    orig_sync = torch.cuda.synchronize
    state = {"n": 0}

    def sync():
        orig_sync()
        state["n"] += 1
        if state["n"] == 1:
            prof.__enter__()
        elif state["n"] == 2:
            prof.__exit__(None, None, None)
    torch.cuda.synchronize = sync
    B.measure(argv + ["--steps", "5", "--warm", "3"], quiet=True)
    torch.cuda.synchronize = orig_sync
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))


if __name__ == "__main__":
    main()
