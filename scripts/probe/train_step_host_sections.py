#!/usr/bin/env python3
"""Host time of a training step by section (perf_counter around the calls, no profiler): where does the host spend / block?
    python scripts/probe/train_step_host_sections.py [--finetune] [--conf-shape]"""
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gens_amd import ops  # noqa: E402
from gens_amd.models.modules import implicit_surface as IS  # noqa: E402
from gens_amd.models.modules import projector  # noqa: E402
from scripts import train_step_bench as B  # noqa: E402

acc = collections.defaultdict(list)


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label].append(time.perf_counter() - t)
    setattr(obj, name, wrapper)


timed(IS.ImplicitSurface, "render")
timed(IS.ImplicitSurface, "_sample_rays")
timed(IS.ImplicitSurface, "_render_core_train")
timed(IS.ImplicitSurface, "_host_draws")
timed(IS.ImplicitSurface, "tv_regularization")
timed(IS, "Scene")
timed(IS, "surface_patch_warp")
timed(ops, "StepPoints")
timed(ops, "blend_train")
timed(ops, "composite")
timed(ops.SdfTrainStep, "__call__", "K17 fwd")
timed(ops.SdfTrainStep, "first_order")
timed(torch.Tensor, "backward")
timed(torch.optim.Adam, "step", "adam")
timed(torch.optim.Adam, "zero_grad")
flags = [a for a in sys.argv[1:] if a.startswith("--")]
ms, label = B.measure(flags + ["--steps", "40", "--warm", "5"], quiet=True)
n = max(len(acc["render"]), 1)
print(f"{label}: {ms:.2f} ms/step; host milliseconds per call by section, MEDIAN over the steps (nested sections overlap):")
med = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
for k, v in sorted(med.items(), key=lambda kv: -kv[1]):
    print(f"  {k:24s} {v * 1e3:7.3f} ms   x{len(acc[k]) / n:.1f}")
