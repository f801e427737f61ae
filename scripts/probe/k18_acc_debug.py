"""Debug: the eleven [dW | db] blocks of K18's backward from the in-kernel sums against operand rows + K14."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_hip_blend import _setup  # noqa: E402
from gens_amd import synthetic  # noqa: E402
from gens_amd.ops import base  # noqa: E402

nv, nl, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
res = {}
for mode in ("rows", "inside"):
    ops, net, views, pts = _setup(nv, nl, seed=50 + nv, n=n)
    base.kernels.blend_train_wgrad = mode
    sc = synthetic.make_scene(nv=nv, h=48, w=64, n_levels=nl, seed=50 + nv)
    views = ops.SceneViews(sc["imgs"].cuda(), sc["intrs"].cuda(), sc["c2ws"].cuda(), [f.cuda() for f in sc["features"]])
    g = torch.Generator().manual_seed(3)
    cot = torch.randn(n, 3, generator=g).cuda()
    rgb, vis = ops.blend_train(net, views, pts)
    (rgb * cot).sum().backward()
    torch.cuda.synchronize()
    res[mode] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
ops, net, views, pts = _setup(nv, nl, seed=50 + nv, n=n)          # the PyTorch layers on K4's look-up as a third opinion
sc = synthetic.make_scene(nv=nv, h=48, w=64, n_levels=nl, seed=50 + nv)
views = ops.SceneViews(sc["imgs"].cuda(), sc["intrs"].cuda(), sc["c2ws"].cuda(), [f.cuda() for f in sc["features"]])
fv, rd, mk = ops.lookup_feature(pts, views)
cot = torch.randn(n, 3, generator=torch.Generator().manual_seed(3)).cuda()
(net(fv, rd, mk) * cot).sum().backward()
res["torch"] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
for k in res["rows"]:
    t = res["torch"][k]
    print(f"{k:22s} torch max {float(t.abs().max()):.3e}  rows - torch {float((res['rows'][k] - t).abs().max()):.3e}  inside - torch {float((res['inside'][k] - t).abs().max()):.3e}")
for k in []:
    a, b = res["rows"][k], res["inside"][k]
    print(f"{k:22s} max |rows| {float(a.abs().max()):.3e}  max diff {float((a - b).abs().max()):.3e}  finite {bool(torch.isfinite(b).all())}")
