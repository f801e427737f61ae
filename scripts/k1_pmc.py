#!/usr/bin/env python3
"""K1 (D=256) a few times, for `rocprofv3 --pmc ... -- python3 scripts/k1_pmc.py` (counters per dispatch: scripts/pmc_summary.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
intrs, c2ws = sc["intrs"].to(dev), sc["c2ws"].to(dev)
w2c = torch.linalg.inv(c2ws).contiguous()
d = 256
tex = ops.pack_nchw(sc["features"][0].to(dev))
nv, h, w, _ = tex.shape
vol, mask = torch.empty(8, d, d, d, device=dev), torch.empty(d, d, d, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    L.call("gens_volume_build_fwd", L.ptr(tex), L.ptr(w2c), L.ptr(intrs), 1.0, nv, h, w, d, 1, L.ptr(vol), L.ptr(mask), L.stream())
torch.cuda.synchronize()
