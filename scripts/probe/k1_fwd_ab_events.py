"""K1 forward at 256^3, timed launch by launch with HIP events (scripts/kernel_bench.py's protocol), with and without the empty-tile shortcut."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
feats = [f.to(dev) for f in sc["features"]]
intrs, c2ws = sc["intrs"].to(dev), sc["c2ws"].to(dev)
w2c = torch.linalg.inv(c2ws).contiguous()
L.load()
d = 256
tex = ops.pack_nchw(feats[0])
nv, h, w, _ = tex.shape
vol, mask = torch.empty(8, d, d, d, device=dev), torch.empty(d, d, d, device=dev)


def run():
    L.call("gens_volume_build_fwd", L.ptr(tex), L.ptr(w2c), L.ptr(intrs), 1.0, nv, h, w, d, 1, L.ptr(vol), L.ptr(mask), L.stream(), nbytes=1)


for _ in range(300):
    run()
torch.cuda.synchronize()
a = nv * h * w * 16 + 36 * d ** 3
for rep in range(3):
    for name in ("shortcut", "general"):
        if name == "general":
            os.environ["GENS_K1_NO_EMPTY_SHORTCUT"] = "1"
        else:
            os.environ.pop("GENS_K1_NO_EMPTY_SHORTCUT", None)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        L.profile_begin()
        for _ in range(50):
            run()
        rec = L.profile_end(raw=True)
        t = [ms * 1e3 for _, ms, _, _ in rec]
        med = statistics.median(t)
        print(f"{name:9s} median {med:7.1f} us  p10 {sorted(t)[5]:7.1f}  {a / med / 1e3 / 80:.1f} % of 8 TB/s")
