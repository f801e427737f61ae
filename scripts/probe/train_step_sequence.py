#!/usr/bin/env python3
"""The launches of ONE training step in issue order, each with the Python line that caused it (torch.profiler with stacks over one
step of scripts/train_step_bench.py): the work list for taking the torch glue out of a step.

    python scripts/probe/train_step_sequence.py [--finetune] [--out FILE]
"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
flags = [a for a in sys.argv[1:] if a in ("--finetune", "--full", "--levels5")]
sys.argv = [sys.argv[0], *flags]
spec_path = os.path.join(ROOT, "scripts", "train_step_bench.py")
src = open(spec_path).read().replace('if __name__ == "__main__":\n    main()', "")
src = src.replace("    for _ in range(int(sys.argv[sys.argv.index(\"--warm\") + 1]) if \"--warm\" in sys.argv else 2):\n        float(step())\n",
                  "    for _ in range(3):\n        float(step())\n    import builtins\n    builtins._gens_step = step\n    return\n")
ns = {"__name__": "tsb", "__file__": spec_path}
exec(compile(src, spec_path, "exec"), ns)
ns["main"]()
import builtins  # noqa: E402

step = builtins._gens_step
torch.cuda.synchronize()
from gens_amd import lib as L  # noqa: E402

_call = L.call


def traced_call(name, *a, **k):          # the C-ABI launches appear in the list under their entry-point names
    with torch.profiler.record_function("ABI:" + (k.get("label") or name)):
        return _call(name, *a, **k)


L.call = traced_call
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    float(step())
    torch.cuda.synchronize()

events = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
kern_of = {}
lines = []
n_kernels = 0


def frame(e):
    for s in e.stack or []:
        if ("gens_amd/" in s or "scripts/" in s) and "lib.py" not in s:
            return s.replace(ROOT + "/", "")
    return (e.stack or ["?"])[0].replace(ROOT + "/", "")


def device_kernels(e):
    ks = list(e.kernels)
    for c in e.cpu_children:
        ks += device_kernels(c)
    return ks


tops = [e for e in events if e.cpu_parent is None]
tops.sort(key=lambda e: e.time_range.start)
for e in tops:
    ks = device_kernels(e)
    if not ks:
        continue
    n_kernels += len(ks)
    dur = sum(k.duration for k in ks)
    shapes = str(e.input_shapes)[:70] if e.input_shapes else ""
    lines.append(f"{len(ks):3d} {dur:8.1f}us  {e.name[:44]:44s} {shapes:70s} {frame(e)[:90]}")
text = "\n".join(lines) + f"\n# {n_kernels} device launches in the step ({' '.join(flags) or 'hot path'})\n"
if out_path:
    open(out_path, "w").write(text)
print(text[-6000:])
