#!/usr/bin/env python3
"""surface_split variant of auto_graph_probe.py in THIS process (for AMD_LOG_LEVEL tracing of the crashing call)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_hip_auto_graph import _finetune_model, _step_inputs  # noqa: E402
from tests.test_hip_ddp import _loss  # noqa: E402

model = _finetune_model(False)
surf = model.implicit_surface
surf.auto_graph = False
ipts = _step_inputs(0)


def fwd():
    return model("train", ipts, cos_anneal_ratio=1.0)


for _ in range(2):
    _loss(fwd(), ipts).backward()
torch.cuda.synchronize()
surf.begin_capture()
f = torch.cuda.CUDAGraph()
with torch.cuda.graph(f):
    out = fwd()
surf.end_capture()
print("forward captured", flush=True)
sys.stderr.write("=== FORWARD CAPTURED ===\n")
ys = [out["color_fine"]]
gos = [torch.zeros_like(y) for y in ys]
params = [p for p in model.parameters() if p.requires_grad]
torch.cuda.synchronize()
b = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    b.capture_begin(pool=f.pool())
    sys.stderr.write("=== BWD CAPTURE BEGUN ===\n")
    gi = torch.autograd.grad(ys, params, gos, retain_graph=True, allow_unused=True)
    sys.stderr.write("=== BWD OPS ISSUED ===\n")
    b.capture_end()
print("backward captured", flush=True)
f.replay()
b.replay()
torch.cuda.synchronize()
print("ok")
