#!/usr/bin/env python3
"""K1 forward under the step's conditions (cold L2 / Infinity Cache: a 1 GiB fill before every launch): the order in which workgroups take their chunks.
GENS_K1_INTERLEAVE=G gives consecutive workgroups chunks from G different parts of the volume.  The variants alternate launch by launch, so that a
drifting clock or a noisy box moves all of them alike."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L  # noqa: E402
from gens_amd import ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
dims = [256, 128, 64]
seed = int(os.environ.get("AB_SEED", "0"))
nv = int(os.environ.get("AB_NV", "5"))
dist = float(os.environ.get("AB_DIST", "2.2"))
sc = synthetic.make_scene(nv=nv, h=480, w=640, n_levels=5, seed=seed, dist=dist)
feats = [f.to(dev) for f in sc["features"]]
intrs, c2ws = sc["intrs"].to(dev), sc["c2ws"].to(dev)
trash = torch.empty(1 << 28, device=dev)
variants = [int(a) for a in sys.argv[1:]] or [0, 16, 32, 64, 128, 256]
algo = nv * 480 * 640 * 16 * (1 + 0.25 + 0.0625) + 36 * sum(d ** 3 for d in dims)
with torch.no_grad():
    ref = ops.volume_build(feats, intrs, c2ws, dims)
    ref = [[t.clone() for t in part] for part in ref]
times = {g: [] for g in variants}
for rep in range(40):
    for g in variants:
        os.environ["GENS_K1_INTERLEAVE"] = str(g)
        trash.fill_(1.0)
        L.profile_begin(only={"gens_volume_build_levels"})
        with torch.no_grad():
            out = ops.volume_build(feats, intrs, c2ws, dims)
        rec = L.profile_end(raw=True)
        times[g].append(sum(ms for _, ms, _, _ in rec) * 1e3)
        if rep == 0:
            same = all(torch.equal(a, b) for pa, pb in zip(out, ref) for a, b in zip(pa, pb))
            print(f"G = {g}: outputs bit-identical to the plain order: {same}")
print(f"scene seed {seed}, {nv} views, cameras at distance {dist}; empty-tile share by level:", [round(float((m == 0).float().mean()), 3) for m in ref[1]])
for g in variants:
    t = sorted(times[g])
    print(f"G = {g:5d}: median {t[len(t) // 2]:6.1f} us  p10 {t[len(t) // 10]:6.1f}  p90 {t[9 * len(t) // 10]:6.1f}   {algo / (t[len(t) // 2] * 1e-6) / 8e12 * 100:.1f} % of 8 TB/s")
