#!/usr/bin/env python3
"""Where a full GenS training step spends its time outside the hot path: the 2-D feature CNN and the 3-D U-Net (PyTorch / MIOpen),
forward and backward, at the benchmark shape (5 views 480x640, cost volumes 256/128/64 with 8 channels)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd.config import gens_model_conf  # noqa: E402
from gens_amd.models.modules.feature_network import FeatureNetwork  # noqa: E402
from gens_amd.models.modules.reg_network import RegNetwork  # noqa: E402

dev = torch.device("cuda:0")
if "--find" in sys.argv:                       # MIOpen's exhaustive "find" instead of the immediate-mode heuristic
    torch.backends.cudnn.benchmark = True
conf = gens_model_conf(volume_dims=(256, 128, 64))
torch.manual_seed(0)


def timed(name, fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e3:9.1f} ms", flush=True)


fnet = FeatureNetwork(conf["feature_network"]).to(dev).train()
imgs = torch.rand(5, 3, 480, 640, device=dev)
timed("feature CNN fwd", lambda: fnet(imgs))
timed("feature CNN fwd+bwd", lambda: sum(o.sum() for o in fnet(imgs)).backward())

rnet = RegNetwork(conf["reg_network"]).to(dev).train()
vols = [torch.randn(1, 8, d, d, d, device=dev, requires_grad=True) for d in (256, 128, 64)]
timed("3-D U-Net fwd", lambda: rnet(vols))
timed("3-D U-Net fwd+bwd", lambda: sum(o.sum() for o in rnet(vols)).backward())

x = vols[0].detach()
for name, m in (("conv3d 8->8 s1 256^3", rnet.conv0.conv), ("conv3d 8->8 s2 256^3", rnet.encoder_layers[0][0].conv),
                ("deconv3d 8->8 128^3->256^3", rnet.decoder_layers[0].conv), ("conv3d 8->4 +bias 256^3", rnet.out_layers[0])):
    inp = x if "deconv" not in name else torch.randn(1, 8, 128, 128, 128, device=dev)
    inp = inp.clone().requires_grad_(True)
    timed(name + " fwd", lambda: m(inp))
    timed(name + " fwd+bwd", lambda: m(inp).sum().backward())
    out = m(inp)
    go = torch.ones_like(out)
    timed(name + " dgrad only", lambda: torch.autograd.grad(out, inp, go, retain_graph=True))
    timed(name + " wgrad only", lambda: torch.autograd.grad(out, m.weight, go, retain_graph=True))
import torch.nn.functional as F  # noqa: E402
wt = rnet.conv0.conv.weight.detach()
timed("torch (MIOpen) conv3d 8->8 s1 256^3 fwd", lambda: F.conv3d(x, wt, None, 1, 1))
from gens_amd import ops  # noqa: E402
xr0 = x.clone().requires_grad_(True)
timed("K16 instnorm+relu 8ch 256^3 fwd", lambda: ops.instnorm_relu(xr0))
timed("K16 instnorm+relu 8ch 256^3 fwd+bwd", lambda: ops.instnorm_relu(xr0).sum().backward())
inorm = torch.nn.InstanceNorm3d(8)
xr = x.clone().requires_grad_(True)
timed("InstanceNorm3d 8ch 256^3 fwd", lambda: inorm(xr))
timed("InstanceNorm3d 8ch 256^3 fwd+bwd", lambda: inorm(xr).sum().backward())
