#!/usr/bin/env python3
"""(Needs the GENS_K1_NO_XCD_ORDER switch of the experiment described in profiles/r05_k1_xcd_order_ab.txt -- measured, not kept; kept for the record.)
K1 forward (all levels, one launch): XCD-contiguous chunk order against the plain order (GENS_K1_NO_XCD_ORDER), interleaved on one box, warm and after a
1 GiB fill (the state the bench's step finds: the render's volumes went through L2 / MALL in between).  Also checks that the outputs are bit-identical."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L  # noqa: E402
from gens_amd import ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
dims = [256, 128, 64]
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
feats = [f.to(dev) for f in sc["features"]]
intrs, c2ws = sc["intrs"].to(dev), sc["c2ws"].to(dev)
trash = torch.empty(1 << 28, device=dev)


def k1(plain):
    if plain:
        os.environ["GENS_K1_NO_XCD_ORDER"] = "1"
    else:
        os.environ.pop("GENS_K1_NO_XCD_ORDER", None)
    with torch.no_grad():
        return ops.volume_build(feats, intrs, c2ws, dims)


a, b = k1(False), k1(True)
for va, vb in zip(a[0] + a[1], b[0] + b[1]):
    assert torch.equal(va, vb)
print("outputs bit-identical")
algo = 5 * 480 * 640 * 16 * (1 + 0.25 + 0.0625) + 36 * sum(d ** 3 for d in dims)
for cold in (False, True):
    ts = {False: [], True: []}
    for it in range(60):
        for plain in (False, True):
            if cold:
                trash.fill_(1.0)
            L.profile_begin(only={"gens_volume_build_levels"})
            held = k1(plain)
            rec = L.profile_end(raw=True)
            ts[plain].append(sum(ms for _, ms, _, _ in rec) * 1e3)
            del held
    for plain in (True, False):
        v = sorted(ts[plain])
        med = v[len(v) // 2]
        print("%-22s %-14s median %6.1f us  p10 %6.1f  p90 %6.1f   %.1f %% of 8 TB/s" % ("after a 1 GiB fill" if cold else "back to back", "plain order" if plain else "XCD order",
                                                                                         med, v[len(v) // 10], v[9 * len(v) // 10], algo / (med * 1e-6) / 8e12 * 100))
