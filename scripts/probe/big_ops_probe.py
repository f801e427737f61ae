#!/usr/bin/env python3
"""Largest torch ops (by GPU time, with shapes) of one fine-tune step: where PyTorch's own kernels are the slow ones."""
import os, sys, builtins, torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.getcwd())
src = open("scripts/train_step_bench.py").read().replace('if __name__ == "__main__":\n    main()', "")
src = src.replace("    for _ in range(2):\n        step()\n", "    for _ in range(2):\n        step()\n    import builtins\n    builtins._gens_step = step\n    return\n")
ns = {"__name__": "tsb", "__file__": os.path.join(os.getcwd(), "scripts", "train_step_bench.py")}
sys.argv = ["x", "--finetune"]
exec(compile(src, "tsb", "exec"), ns)
ns["main"]()
step = builtins._gens_step
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and getattr(e, "device_time_total", 0) > 0]
rows.sort(key=lambda e: -getattr(e, "device_time_total", 0))
for e in rows[:40]:
    print(f"{e.key:12s} n={e.count:4d} gpu {getattr(e,'device_time_total',0)/1e3:7.3f} ms  {str(e.input_shapes)[:110]}")
