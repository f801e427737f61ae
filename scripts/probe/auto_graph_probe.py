#!/usr/bin/env python3
"""Which part of a split forward / backward capture trips the HIP runtime?  Each variant runs in a child process (a segfault must not end the probe)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VARIANTS = {
    "torch_linear_mgc": r'''
import torch
m = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8)).cuda()
x = torch.randn(32, 64, device="cuda")
g = torch.cuda.make_graphed_callables(m, (x,))
y = g(x); y.sum().backward(); torch.cuda.synchronize(); print("ok", float(y.sum()))
''',
    "split_torch_only": r'''
import torch
m = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8)).cuda()
x = torch.randn(32, 64, device="cuda")
for _ in range(2): m(x).sum().backward()
torch.cuda.synchronize()
f = torch.cuda.CUDAGraph()
with torch.cuda.graph(f):
    y = m(x)
go = torch.zeros_like(y)
b = torch.cuda.CUDAGraph()
with torch.cuda.graph(b, pool=f.pool()):
    gi = torch.autograd.grad([y], list(m.parameters()), [go], retain_graph=True)
f.replay(); b.replay(); torch.cuda.synchronize(); print("ok")
''',
    "surface_split": r'''
import sys, torch
sys.path.insert(0, %(root)r)
from tests.test_hip_auto_graph import _finetune_model, _step_inputs
from tests.test_hip_ddp import _loss
MODE = %(mode)r
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
names = [k for k, v in out.items() if torch.is_tensor(v) and v.requires_grad]
if MODE == "color_only":
    names = ["color_fine"]
ys = [out[k] for k in names]
gos = [torch.zeros_like(y) for y in ys]
params = [p for p in model.parameters() if p.requires_grad]
torch.cuda.synchronize()
b = torch.cuda.CUDAGraph()
kw = {} if MODE == "own_pool" else {"pool": f.pool()}
if MODE == "thread_local":
    kw["capture_error_mode"] = "thread_local"
with torch.cuda.graph(b, **kw):
    gi = torch.autograd.grad(ys, params, gos, retain_graph=(MODE != "no_retain"), allow_unused=True)
print("backward captured", flush=True)
f.replay(); b.replay(); torch.cuda.synchronize(); print("ok")
''',
}


def main():
    runs = [("torch_linear_mgc", {}), ("split_torch_only", {})] + [("surface_split", {"mode": m}) for m in ("shared_pool", "own_pool", "thread_local", "no_retain", "color_only")]
    for name, kw in runs:
        code = VARIANTS[name] % dict(kw, root=ROOT) if kw or "%(" in VARIANTS[name] else VARIANTS[name]
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
        tail = (r.stdout.strip().splitlines() or [""])[-1]
        err = [ln for ln in r.stderr.splitlines() if "Error" in ln or "error" in ln or "Fatal" in ln][:3]
        print(f"{name} {kw}: rc={r.returncode} last='{tail}' {err}", flush=True)


if __name__ == "__main__":
    main()
