#!/usr/bin/env python3
"""K1 backward on the image-tile plan (gens_volume_build_bwd_levels: all levels in five launches, means and counts from the forward pass) against the
wave-window kernel and against direct device atomics: time and agreement."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
nv, h0, w0 = (3, 1152, 1600) if "--conf-shape" in sys.argv else (5, 480, 640)
dims = [256, 128, 64, 32, 16] if "--levels5" in sys.argv or "--conf-shape" in sys.argv else [256, 128, 64]
n = len(dims)
sc = synthetic.make_scene(nv=nv, h=h0, w=w0, n_levels=5, seed=0)
cams = ops.SceneCams.of(sc["intrs"].to(dev), sc["c2ws"].to(dev))
texs = [ops.pack_nchw(sc["features"][l].to(dev)) for l in range(n)]
hw = [x for t in texs for x in (t.shape[1], t.shape[2])]
vols = [torch.empty(8, d, d, d, device=dev) for d in dims]
masks = [torch.empty(d, d, d, device=dev) for d in dims]
counts = [torch.empty(d ** 3, device=dev, dtype=torch.uint8) for d in dims]
intrs = [cams.ks[l] for l in range(n)]
L.call("gens_volume_build_levels", L.ptr_table(texs), L.int_table(hw), L.int_table(dims), n, L.ptr(cams.w2c), L.ptr_table(intrs), nv, 1,
       L.ptr_table(vols), L.ptr_table(masks), L.ptr_table(counts, torch.uint8), L.stream())
print("visible views per voxel, by level:", [round(float(c.float().mean()), 3) for c in counts])
g = torch.Generator(device=dev).manual_seed(1)
gvols = [torch.randn(8, d, d, d, device=dev, generator=g) for d in dims]
need = L.load().gens_volume_build_bwd_levels_scratch_bytes(L.int_table(hw), L.int_table(dims), n, nv)
scratch = torch.empty(need, device=dev, dtype=torch.uint8)
print(f"scratch {need / 2**20:.1f} MiB")


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def levels(on, out):
    gv = [gvols[l] if on[l] else None for l in range(n)]
    L.call("gens_volume_build_bwd_levels", L.ptr_table(texs), L.int_table(hw), L.int_table(dims), n, L.ptr(cams.w2c), L.ptr_table(intrs), nv, L.ptr_table(vols),
           L.ptr_table(counts, torch.uint8), L.ptr_table(gv), L.ptr_table(out), L.ptr(scratch, torch.uint8), need, L.stream())


if "--lists" in sys.argv:         # what the plan produced (the scratch layout of bwd_levels_layout, k1_volume.hip): pairs, work items, neighbours in the lists
    import numpy as np
    levels([True] * n, [torch.zeros_like(t) for t in texs])
    torch.cuda.synchronize()
    raw = scratch.cpu().numpy()
    al = lambda v: (v + 255) // 256 * 256  # noqa: E731
    tiles = [((hw[2 * l + 1] + 63) // 64) * ((hw[2 * l] + 59) // 60) for l in range(n)]
    bins = sum(nv * t + 1 for t in tiles)
    nws = [d ** 3 // 64 for d in dims]
    segs = [min(512, max(64, nw * nv // 512 // 64 * 64)) for nw in nws]
    items_cap = sum((nw * nv * 4 + sg - 1) // sg + nv * t + (nw * nv + 15) // 16 + 1 for nw, sg, t in zip(nws, segs, tiles))
    at = 0
    count = raw[at:at + bins * 4].view(np.uint32); at += al(bins * 4)
    at += al(bins * 4)
    n_items = int(raw[at:at + 4].view(np.uint32)[0]); at += al(4)
    at += al(4)
    offset = raw[at:at + (bins + 1) * 4].view(np.uint32); at += al((bins + 1) * 4)
    items = raw[at:at + items_cap * 16].view(np.uint32).reshape(-1, 4)[:n_items]; at += al(items_cap * 16)
    lst = raw[at:at + int(offset[bins]) * 4].view(np.uint32)
    print(f"bins {bins}, entries {int(offset[bins])} for {sum(nws) * nv} pairs, work items {n_items}")
    b0 = 0
    for l in range(n):
        nb = nv * tiles[l]
        c = count[b0:b0 + nb + 1]
        ent = int(c[:nb].sum())
        lo, hi = int(offset[b0]), int(offset[b0 + nb])
        it = items[items[:, 3] == l]
        sizes = it[:, 2] - it[:, 1]
        print(f"level {l} (D={dims[l]}): tile entries {ent} ({ent / (nws[l] * nv):.2f} per pair), direct pairs {int(c[nb])}, non-empty bins {int((c[:nb] > 0).sum())} of {nb}, "
              f"largest bin {int(c[:nb].max())}; items {len(it)}, median size {int(np.median(sizes)) if len(it) else 0}")
        b0 += nb + 1
    sys.exit(0)
if "--masks" in sys.argv:         # debug: which combinations of levels run
    scr = [torch.zeros_like(t) for t in texs]
    for on in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1], [1, 1, 1]):
        print(on, flush=True)
        levels([bool(x) for x in on] + [False] * (n - 3), scr)
        torch.cuda.synchronize()
        print("ok", flush=True)
    sys.exit(0)
if "--counters" in sys.argv:      # workload for scripts/pmc_counters.sh: the 256^3 level alone, five calls
    scr = [torch.zeros_like(t) for t in texs]
    for _ in range(5):
        levels([True] + [False] * (n - 1), scr)
    torch.cuda.synchronize()
    sys.exit(0)
outs = [torch.zeros_like(t) for t in texs]
levels([True] * n, outs)
torch.cuda.synchronize()
for l, d in enumerate(dims):
    ref = torch.zeros_like(texs[l])
    os.environ["GENS_K1_BWD_DIRECT"] = "1"
    L.call("gens_volume_build_bwd", L.ptr(texs[l]), L.ptr(cams.w2c), L.ptr(intrs[l]), 1.0, nv, hw[2 * l], hw[2 * l + 1], d, L.ptr(gvols[l]), L.ptr(ref), L.stream())
    os.environ.pop("GENS_K1_BWD_DIRECT")
    torch.cuda.synchronize()
    print(f"D={d}: max |levels - direct| = {float((outs[l] - ref).abs().max()):.3e}   (max |g| = {float(ref.abs().max()):.3e})")
scr = [torch.zeros_like(t) for t in texs]
print(f"all levels, one launch set: {timed(lambda: levels([True] * n, scr)):.3f} ms")
for l, d in enumerate(dims):
    on = [k == l for k in range(n)]
    t3 = timed(lambda: levels(on, scr))
    tw = timed(lambda: L.call("gens_volume_build_bwd", L.ptr(texs[l]), L.ptr(cams.w2c), L.ptr(intrs[l]), 1.0, nv, hw[2 * l], hw[2 * l + 1], d, L.ptr(gvols[l]),
                              L.ptr(scr[l]), L.stream()))
    print(f"D={d}: levels (this level only) {t3:.3f} ms   wave-window kernel {tw:.3f} ms")
