"""K18 forward / backward launch times inside a hot-path training step (HIP events per C-ABI launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scripts.train_step_bench import measure  # noqa: E402

ms, label, kt = measure(["--steps", "10", "--warm", "3"], quiet=True, kernels=True)
for name, k in sorted(kt.items(), key=lambda kv: -kv[1]["ms"])[:12]:
    print(f"{name:36s} {k['launches']:3d} launches  {k['ms']:8.3f} ms per step")
print(label, f"{ms:.2f} ms")
