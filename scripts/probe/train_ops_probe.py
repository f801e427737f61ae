#!/usr/bin/env python3
"""Which torch ops make up the ~3 000 launches of a training step?  torch.profiler over one step of scripts/train_step_bench.py's loop."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0]]
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("tsb", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "train_step_bench.py"))
tsb = importlib.util.module_from_spec(spec)
src = open(spec.origin).read().replace('if __name__ == "__main__":\n    main()', "")
src = src.replace("    for _ in range(2):\n        step()\n", "    for _ in range(2):\n        step()\n    import builtins\n    builtins._gens_step = step\n    return\n")
exec(compile(src, spec.origin, "exec"), tsb.__dict__)
tsb.main()
import builtins  # noqa: E402

step = builtins._gens_step
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.count > 0]
rows.sort(key=lambda e: -e.count)
print(f"{'op':60s} {'count':>6s} {'cpu ms':>8s} {'gpu ms':>8s}")
for e in rows[:45]:
    print(f"{e.key[:60]:60s} {e.count:6d} {e.cpu_time_total / 1e3:8.2f} {getattr(e, 'device_time_total', 0) / 1e3:8.2f}")
