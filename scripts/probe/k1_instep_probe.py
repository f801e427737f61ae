#!/usr/bin/env python3
"""Why does gens_volume_build_levels take 0.30 - 0.33 ms inside a bench step and 0.25 ms in an isolated loop?  HIP-event time of the one launch
under the conditions a step adds, one at a time: an idle GPU before it (the step starts after the previous image's read-back), cold caches (the
render's 300 MB of volumes went through L2 / MALL in between), freshly allocated outputs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L  # noqa: E402
from gens_amd import ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
dims = [256, 128, 64]
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
feats = [f.to(dev) for f in sc["features"]]
intrs, c2ws = sc["intrs"].to(dev), sc["c2ws"].to(dev)
trash = torch.empty(1 << 28, device=dev)          # 1 GiB


def k1():
    with torch.no_grad():
        return ops.volume_build(feats, intrs, c2ws, dims)


def timed(prepare, n=40, keep=False):
    ts = []
    held = None
    for _ in range(n):
        if not keep:
            held = None
        prepare()
        L.profile_begin(only={"gens_volume_build_levels"})
        held = k1()
        rec = L.profile_end(raw=True)
        ts.append(sum(ms for name, ms, _, _ in rec))
    ts.sort()
    return ts[len(ts) // 2] * 1e3, ts[len(ts) // 10] * 1e3, ts[9 * len(ts) // 10] * 1e3


def nothing():
    pass


def idle(ms):
    def f():
        torch.cuda.synchronize()
        time.sleep(ms / 1e3)
    return f


def thrash():
    trash.fill_(1.0)


def thrash_then_touch_texels():
    """cold caches, then ONE pass over the packed texels K1 reads (33 MB: ~8 us): is the cold penalty the texel reads?"""
    trash.fill_(1.0)
    from gens_amd.ops.base import pack_maps
    for t in pack_maps(feats[:len(dims)]):
        t.view(-1)[::16].sum()                       # (one float per 64-byte line)


def thrash_then_idle():
    trash.fill_(1.0)
    torch.cuda.synchronize()
    time.sleep(0.002)


for _ in range(5):
    k1()
torch.cuda.synchronize()
algo = 5 * 480 * 640 * 16 * (1 + 0.25 + 0.0625) + 36 * sum(d ** 3 for d in dims)
for name, prep in (("back to back", nothing), ("after 0.2 ms idle", idle(0.2)), ("after 2 ms idle", idle(2)), ("after 20 ms idle", idle(20)),
                   ("after a 1 GiB fill (cold L2 / MALL), no idle", thrash), ("after the fill and 2 ms idle", thrash_then_idle),
                   ("after the fill and a pass over the texels", thrash_then_touch_texels)):
    med, p10, p90 = timed(prep)
    print("%-48s median %6.1f us  p10 %6.1f  p90 %6.1f   %.1f %% of 8 TB/s" % (name, med, p10, p90, algo / (med * 1e-6) / 8e12 * 100))

for g in (8, 64, 512, 4096):
    os.environ["GENS_K1_INTERLEAVE"] = str(g)
    print(f"--- consecutive workgroups from {g} different parts of the volume (GENS_K1_INTERLEAVE={g})")
    for name, prep in (("back to back", nothing), ("after a 1 GiB fill (cold L2 / MALL), no idle", thrash)):
        med, p10, p90 = timed(prep)
        print("%-48s median %6.1f us  p10 %6.1f  p90 %6.1f   %.1f %% of 8 TB/s" % (name, med, p10, p90, algo / (med * 1e-6) / 8e12 * 100))
os.environ.pop("GENS_K1_INTERLEAVE")
os.environ["GENS_K1_WARM"] = "0"
print("--- the same with the launch's own texel warm-up switched off (GENS_K1_WARM=0)")
for name, prep in (("back to back", nothing), ("after 0.2 ms idle", idle(0.2)), ("after a 1 GiB fill (cold L2 / MALL), no idle", thrash),
                   ("after the fill and a pass over the texels", thrash_then_touch_texels)):
    med, p10, p90 = timed(prep)
    print("%-48s median %6.1f us  p10 %6.1f  p90 %6.1f   %.1f %% of 8 TB/s" % (name, med, p10, p90, algo / (med * 1e-6) / 8e12 * 100))
os.environ["GENS_K1_WARM"] = "2"
print("--- the warm-up as a launch of its own in front (GENS_K1_WARM=2; both launches inside the timed entry point)")
for name, prep in (("back to back", nothing), ("after 0.2 ms idle", idle(0.2)), ("after a 1 GiB fill (cold L2 / MALL), no idle", thrash),
                   ("after the fill and a pass over the texels", thrash_then_touch_texels)):
    med, p10, p90 = timed(prep)
    print("%-48s median %6.1f us  p10 %6.1f  p90 %6.1f   %.1f %% of 8 TB/s" % (name, med, p10, p90, algo / (med * 1e-6) / 8e12 * 100))
os.environ.pop("GENS_K1_WARM")
print("--- one launch for all levels against level by level (GENS_K1_PER_LEVEL), outputs kept, 0.2 ms idle before each")
for env in (None, "1"):
    if env is None:
        os.environ.pop("GENS_K1_PER_LEVEL", None)
    else:
        os.environ["GENS_K1_PER_LEVEL"] = env
    for _ in range(3):
        k1()
    med, p10, p90 = timed(idle(0.2), keep=True)
    print("%-48s median %6.1f us  p10 %6.1f  p90 %6.1f   %.1f %% of 8 TB/s" % ("per level" if env else "one launch", med, p10, p90, algo / (med * 1e-6) / 8e12 * 100))
os.environ.pop("GENS_K1_PER_LEVEL", None)
