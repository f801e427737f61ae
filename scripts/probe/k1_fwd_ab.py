"""A/B of the K1 forward's empty-tile shortcut on one box: the same launches with and without GENS_K1_NO_EMPTY_SHORTCUT, interleaved."""
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
for lvl, d in enumerate([256, 128, 64]):
    tex = ops.pack_nchw(feats[lvl])
    nv, h, w, _ = tex.shape
    k = intrs.clone()
    k[:, :2] *= 0.5 ** lvl
    vol, mask = torch.empty(8, d, d, d, device=dev), torch.empty(d, d, d, device=dev)

    def run():
        L.call("gens_volume_build_fwd", L.ptr(tex), L.ptr(w2c), L.ptr(k), 1.0, nv, h, w, d, 1, L.ptr(vol), L.ptr(mask), L.stream())
    for _ in range(200):
        run()
    torch.cuda.synchronize()
    res = {"shortcut": [], "general": []}
    for rep in range(6):
        for name in ("shortcut", "general"):
            if name == "general":
                os.environ["GENS_K1_NO_EMPTY_SHORTCUT"] = "1"
            else:
                os.environ.pop("GENS_K1_NO_EMPTY_SHORTCUT", None)
            run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                run()
            e.record()
            torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) / 50 * 1e3)
    a = nv * h * w * 16 + 36 * d ** 3
    ref = vol.clone()
    os.environ.pop("GENS_K1_NO_EMPTY_SHORTCUT", None)
    run()
    torch.cuda.synchronize()
    same = torch.equal(ref, vol)
    empty = float((mask == 0).float().mean())
    for name in ("shortcut", "general"):
        t = statistics.median(res[name])
        print(f"K1 D={d} {name:9s}: {t:7.1f} us (min {min(res[name]):.1f})  {a / t / 1e3:7.1f} GB/s = {a / t / 1e3 / 80:.1f} % of 8 TB/s")
    print(f"   bit-identical: {same}; voxels with mask 0: {100 * empty:.1f} %")
