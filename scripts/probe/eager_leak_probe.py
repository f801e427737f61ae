#!/usr/bin/env python3
"""Does an EAGER training step leave autograd nodes / device memory behind?  memory_allocated() after every step of the small fine-tune model."""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_hip_auto_graph import _finetune_model, _step_inputs  # noqa: E402
from tests.test_hip_ddp import _loss  # noqa: E402

model = _finetune_model(False)
opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "vol_lr": [1e-2, 1e-2, 1e-2]}))
seen = []
for k in range(12):
    ipts = _step_inputs(k)
    out = model("train", ipts, cos_anneal_ratio=1.0)
    loss = _loss(out, ipts)
    opt.zero_grad()
    loss.backward()
    opt.step()
    float(loss)
    del out, loss, ipts
    if k == 8:
        gc.collect()
    torch.cuda.synchronize()
    n_nodes = sum(1 for o in gc.get_objects() if type(o).__name__.endswith("Backward") and "torch.autograd.function" in str(type(o).__mro__))
    seen.append((torch.cuda.memory_allocated(), n_nodes))
    print("step", k, "allocated", seen[-1][0], "python autograd nodes alive", n_nodes)
from collections import Counter
names = Counter(type(o).__name__ for o in gc.get_objects() if type(o).__name__.endswith("Backward") and "torch.autograd.function" in str(type(o).__mro__))
print("leaked node types:", dict(names))
for o in gc.get_objects():
    if type(o).__name__ in names:
        attrs = {k: type(v).__name__ for k, v in vars(o).items()} if hasattr(o, "__dict__") else {}
        print(type(o).__name__, attrs)
        break
