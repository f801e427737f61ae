"""A/B of the gather kernels' round-4 changes on one box (interleaved, same inputs as scripts/kernel_bench.py):
K2 forward (packed) and K4 forward with / without the XCD-contiguous block order, K4's vectorised copy-out, K3 after the division fix."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gens_amd import lib as L, ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
L.load()
sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
imgs, intrs, c2ws = sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev)
feats = [f.to(dev) for f in sc["features"]]
dims = [256, 128, 64]
vols = [v.to(dev) for v in synthetic.make_volumes(dims, seed=1)]
b = 32768
ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640)
ro, rd = ro[:b].to(dev).contiguous(), rd[:b].to(dev).contiguous()
near, far = sc["near"].to(dev), sc["far"].to(dev)


def timed(fn, reps=40, rounds=5):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / reps * 1e3)
    return statistics.median(out)


def ab(label, fn, env, nbytes):
    res = {}
    for rep in range(2):
        for on in (False, True):
            if on:
                os.environ[env] = "1"
            else:
                os.environ.pop(env, None)
            res.setdefault(on, []).append(timed(fn))
    os.environ.pop(env, None)
    t_new, t_old = min(res[False]), min(res[True])
    print(f"{label:44s} now {t_new:8.1f} us = {nbytes / t_new / 1e3 / 80:5.1f} %   with {env}: {t_old:8.1f} us = {nbytes / t_old / 1e3 / 80:5.1f} % of 8 TB/s")


with torch.no_grad():
    _, masks = ops.volume_build(feats[:3], intrs, c2ws, dims)
    mset = ops.VolumeSet.masks(masks)
    vpack = ops.VolumeSet.packed(vols)
    z64 = (near + (far - near) * torch.linspace(0, 1, 64, device=dev)[None]).expand(b, 64).contiguous()
    z128 = (near + (far - near) * torch.linspace(0, 1, 128, device=dev)[None]).expand(b, 128).contiguous()
    for z, mid in ((z64, False), (z128, True)):
        n = z.shape[1]
        t = timed(lambda: ops.ray_points(ro, rd, z, mset, mid=mid, sample_dist=1 / 32))
        nb = b * n * 17 + b * 24
        print(f"K3 ray_points B={b} n={n:<28d} now {t:8.1f} us = {nb / t / 1e3 / 80:5.1f} % of 8 TB/s")
    pts, valid = ops.ray_points(ro, rd, z128, mset, mid=True, sample_dist=1 / 32)
    npts = pts.shape[0]
    ab(f"K2 lookup fwd packed N={npts} L=3", lambda: ops.lookup_volume(pts, vpack), "GENS_NO_XCD_REMAP", npts * (12 + 16 * 3))
    ab(f"K2 lookup fwd packed N={npts} L=3 (pairs)", lambda: ops.lookup_volume(pts, vpack), "GENS_K2_NO_PAIRS", npts * (12 + 16 * 3))
    os.environ["GENS_K2_NO_PAIRS"] = "1"
    ref = ops.lookup_volume(pts, vpack)
    os.environ.pop("GENS_K2_NO_PAIRS")
    print("K2 paired forward bit-identical to the lane-per-item kernel:", bool(torch.equal(ref, ops.lookup_volume(pts, vpack))))
    views = ops.SceneViews(imgs, intrs, c2ws, feats)
    pv = pts[valid.bool()].contiguous()
    nvp, s = pv.shape[0], 4
    nb4 = nvp * 12 + nvp * s * (4 * (3 + 4 * 5) + 17)
    ab(f"K4 lookup_feature N={nvp} S=4 L_f=5 (remap)", lambda: ops.lookup_feature(pv, views), "GENS_NO_XCD_REMAP", nb4)
    ab(f"K4 lookup_feature N={nvp} S=4 L_f=5 (copy)", lambda: ops.lookup_feature(pv, views), "GENS_K4_PLAIN_COPY", nb4)
    ab(f"K4 lookup_feature N={nvp} S=4 L_f=5 (pairs)", lambda: ops.lookup_feature(pv, views), "GENS_K4_NO_PAIRS", nb4)
    ab(f"K4 lookup_feature N={nvp} S=4 L_f=5 (unroll)", lambda: ops.lookup_feature(pv, views), "GENS_K4_NO_UNROLL", nb4)
    os.environ["GENS_K4_NO_PAIRS"] = "1"
    ref = ops.lookup_feature(pv, views)
    os.environ.pop("GENS_K4_NO_PAIRS")
    new = ops.lookup_feature(pv, views)
    print("K4 paired forward bit-identical to the lane-per-item kernel:", all(bool(torch.equal(a, b)) for a, b in zip(ref, new)))
    for s3 in (3, 2):
        v3 = ops.SceneViews(imgs[:s3 + 1], intrs[:s3 + 1], c2ws[:s3 + 1], [f[:s3 + 1].contiguous() for f in feats])
        pv3 = pv[:1000003]
        os.environ["GENS_K4_NO_PAIRS"] = "1"
        ref = ops.lookup_feature(pv3, v3)
        os.environ.pop("GENS_K4_NO_PAIRS")
        new = ops.lookup_feature(pv3, v3)
        print(f"  S={s3}, odd point count:", all(bool(torch.equal(a, b)) for a, b in zip(ref, new)))
