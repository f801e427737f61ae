#!/usr/bin/env python3
"""Isolated per-kernel measurement of the gather / per-ray kernels (SURVEY.md section 8d timing protocol):
>= 20 warm-up + >= 100 timed launches each (after ~60 ms of work that brings the clocks up), HIP events on the launch stream,
median / p10 / p90 and the algorithmic HBM fraction (A / t against 8 TB/s).  Shapes are those of one bench.py ray chunk (32 768 rays, 5 views, dims 256/128/64).

    python scripts/kernel_bench.py [--out profiles/rNN_kernels_isolated.json] [--iters 100]
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gens_amd import lib as L  # noqa: E402
from gens_amd import ops, synthetic  # noqa: E402

HBM = 8000.0  # GB/s


def measure(fn, iters, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    L.profile_begin()
    for _ in range(iters):
        fn()
    rec = L.profile_end(raw=True)
    per = {}
    for name, ms, _b, _f in rec:
        per.setdefault(name, []).append(ms)
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--rays", type=int, default=32768)
    ap.add_argument("--no-smi", action="store_true", help="skip the rocm-smi snapshot (a child exec is refused under rocprofv3 --pmc)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L.load()
    dims = [256, 128, 64]
    sc = synthetic.make_scene(nv=5, h=480, w=640, n_levels=5, seed=0)
    imgs, intrs, c2ws = sc["imgs"].to(dev), sc["intrs"].to(dev), sc["c2ws"].to(dev)
    feats = [f.to(dev) for f in sc["features"]]
    vols = [v.to(dev) for v in synthetic.make_volumes(dims, seed=100)]
    ro, rd = synthetic.make_rays(sc["intrs"], sc["c2ws"], 480, 640)
    b = args.rays
    sel = slice(150 * 640, 150 * 640 + b)
    ro, rd = ro[sel].to(dev).contiguous(), rd[sel].to(dev).contiguous()
    near, far = sc["near"].to(dev), sc["far"].to(dev)
    w2c = torch.linalg.inv(c2ws).contiguous()
    rows = []

    def add(label, kernel, per, a_bytes, note=""):
        t = np.array(per[kernel]) * 1e3      # us
        med = float(np.median(t))
        rows.append({"case": label, "kernel": kernel, "launches": len(t), "median_us": round(med, 2), "p10_us": round(float(np.percentile(t, 10)), 2),
                     "p90_us": round(float(np.percentile(t, 90)), 2), "algorithmic_MB": round(a_bytes / 1e6, 3),
                     "GBs": round(a_bytes / med / 1e3, 1), "hbm_frac": round(a_bytes / med / 1e3 / HBM, 4), "note": note})
        r = rows[-1]
        print(f"{label:46s} {r['median_us']:9.1f} us (p10 {r['p10_us']:.1f}, p90 {r['p90_us']:.1f})  A={r['algorithmic_MB']:9.2f} MB  "
              f"{r['GBs']:8.1f} GB/s  {100 * r['hbm_frac']:5.1f} %", flush=True)

    with torch.no_grad():
        # clocks up first: the first ~10 ms of work after idle run up to 25 % slower (scripts/probe/README.md)
        tex0 = ops.pack_nchw(feats[0])

        def k1_fwd(tex, k, d, vol, mask):
            nv_, h_, w_, _ = tex.shape
            L.call("gens_volume_build_fwd", L.ptr(tex), L.ptr(w2c), L.ptr(k), 1.0, nv_, h_, w_, d, 1, L.ptr(vol), L.ptr(mask), L.stream(),
                   nbytes=nv_ * h_ * w_ * 16 + 36 * d ** 3)

        vol0, mask0 = torch.empty(8, dims[0], dims[0], dims[0], device=dev), torch.empty(dims[0], dims[0], dims[0], device=dev)
        for _ in range(300):
            k1_fwd(tex0, intrs, dims[0], vol0, mask0)
        torch.cuda.synchronize()
        # ---- K1 per level
        for lvl, d in enumerate(dims):
            tex = ops.pack_nchw(feats[lvl])
            nv, h, w, _ = tex.shape
            k = intrs.clone()
            k[:, :2] = k[:, :2] * 0.5 ** lvl                     # pre-scaled intrinsics, as ops.volume_build passes them
            vol, mask = torch.empty(8, d, d, d, device=dev), torch.empty(d, d, d, device=dev)
            per = measure(lambda: k1_fwd(tex, k, d, vol, mask), max(20, args.iters // 4))
            add(f"K1 volume build D={d} ({nv} views {h}x{w})", "gens_volume_build_fwd", per, nv * h * w * 16 + 36 * d ** 3)
        # ---- K1 backward: all levels in one launch set (the training step's call), from the forward's means and counts
        cams = ops.SceneCams.of(intrs, c2ws)
        texs = [ops.pack_nchw(feats[l]) for l in range(len(dims))]
        hw = [x for t in texs for x in (t.shape[1], t.shape[2])]
        vols_f = [torch.empty(8, d, d, d, device=dev) for d in dims]
        masks_f = [torch.empty(d, d, d, device=dev) for d in dims]
        counts = [torch.empty(d ** 3, device=dev, dtype=torch.uint8) for d in dims]
        ks = [cams.ks[l] for l in range(len(dims))]
        L.call("gens_volume_build_levels", L.ptr_table(texs), L.int_table(hw), L.int_table(dims), len(dims), L.ptr(cams.w2c), L.ptr_table(ks), 5, 1,
               L.ptr_table(vols_f), L.ptr_table(masks_f), L.ptr_table(counts, torch.uint8), L.stream())
        gvols = [torch.randn(8, d, d, d, device=dev) for d in dims]
        gfeat = [torch.zeros_like(t) for t in texs]
        need = L.load().gens_volume_build_bwd_levels_scratch_bytes(L.int_table(hw), L.int_table(dims), len(dims), 5)
        scratch = torch.empty(need, device=dev, dtype=torch.uint8)
        a_bwd = sum(2 * t.numel() * 4 + 32 * d ** 3 for t, d in zip(texs, dims))          # texels read + their gradient written, 8 cotangent planes read
        per = measure(lambda: L.call("gens_volume_build_bwd_levels", L.ptr_table(texs), L.int_table(hw), L.int_table(dims), len(dims), L.ptr(cams.w2c),
                                     L.ptr_table(ks), 5, L.ptr_table(vols_f), L.ptr_table(counts, torch.uint8), L.ptr_table(gvols), L.ptr_table(gfeat),
                                     L.ptr(scratch, torch.uint8), need, L.stream(), nbytes=a_bwd), max(20, args.iters // 4))
        add("K1 backward, levels 256/128/64 in one launch set", "gens_volume_build_bwd_levels", per, a_bwd,
            note="five launches (memset, plan, scan, fill, tiles); what it must read at 1.5 window visits per (voxel tile, view) pair is ~4 x the algorithmic bytes (DESIGN 4e)")
        del vols_f, gvols, scratch
        _, masks = ops.volume_build(feats[:3], intrs, c2ws, dims)
        mset = ops.VolumeSet.masks(masks)
        vpack = ops.VolumeSet.packed(vols)

        # ---- K3 ray points + masks, n = 64 (coarse) and 128 (mid points)
        z64 = (near + (far - near) * torch.linspace(0, 1, 64, device=dev)[None]).expand(b, 64).contiguous()
        z128 = (near + (far - near) * torch.linspace(0, 1, 128, device=dev)[None]).expand(b, 128).contiguous()
        for z, mid in ((z64, False), (z128, True)):
            n = z.shape[1]
            per = measure(lambda: ops.ray_points(ro, rd, z, mset, mid=mid, sample_dist=1 / 32), args.iters)
            add(f"K3 ray_points B={b} n={n}", "gens_ray_points", per, b * n * 17 + b * 24)
        pts, valid = ops.ray_points(ro, rd, z128, mset, mid=True, sample_dist=1 / 32)
        npts = pts.shape[0]

        # ---- K2 look-up (packed texel volumes: the inference layout) and planar fwd
        per = measure(lambda: ops.lookup_volume(pts, vpack), args.iters)
        add(f"K2 lookup fwd packed N={npts} L=3", "gens_lookup_volume_fwd", per, npts * (12 + 16 * 3))

    # ---- K2 backward / second order (planar volumes, training layout) at N = 512*128 (one training step) and at the chunk size
    for n_pts, tag in ((512 * 128, "train step"), (npts // 4, "1/4 chunk")):
        p = pts[:n_pts].clone().requires_grad_(True)
        vs = [v.clone().requires_grad_(True) for v in vols]

        def fwd_bwd():
            f = ops.lookup_volume(p, vs)
            g, = torch.autograd.grad(f.sum(), p, create_graph=True)
            (g * g).sum().backward()
            p.grad = None
            for v in vs:
                v.grad = None
        per = measure(fwd_bwd, max(20, args.iters // 4))
        add(f"K2 lookup fwd planar N={n_pts} ({tag})", "gens_lookup_volume_fwd", per, n_pts * (12 + 16 * 3))
        add(f"K2 lookup bwd N={n_pts} ({tag})", "gens_lookup_volume_bwd", per, n_pts * (24 + 16 * 3), "two launches per step (d/dpts, then with d/dvolume atomics)")
        add(f"K2'' lookup bwd2 N={n_pts} ({tag})", "gens_lookup_volume_bwd2", per, n_pts * (36 + 32 * 3))

    with torch.no_grad():
        # ---- K4 source-view features for the valid points of the chunk
        views = ops.SceneViews(imgs, intrs, c2ws, feats)
        pv = pts[valid.reshape(-1)].contiguous()
        nvp = pv.shape[0]
        s = 4
        per = measure(lambda: ops.lookup_feature(pv, views), max(20, args.iters // 2))
        add(f"K4 lookup_feature N={nvp} S=4 L_f=5", "gens_lookup_feature_fwd", per, nvp * 12 + nvp * s * (4 * (3 + 4 * 5) + 17))

        # ---- K5/K6 up-sample rounds, K7 merge
        sdf = (torch.rand(b, 112, device=dev) - 0.3) * 0.2
        for rnd, n in enumerate((64, 80, 96, 112)):
            zz = z128[:, :n].contiguous()
            ss = sdf[:, :n].contiguous()
            _, vin = ops.ray_points(ro, rd, zz, mset)
            per = measure(lambda: ops.upsample(ro, rd, zz, ss, 16, mset, 64.0 * 2 ** rnd, valid_in=vin), args.iters)
            add(f"K5/K6 upsample B={b} n={n}->16", "gens_upsample", per, b * (9 * n + 17 * 16 + 24), "mask decisions of the n old samples carried in")
        znew = (zz[:, :16] + 1e-3).contiguous()
        snew = ss[:, :16].contiguous()
        per = measure(lambda: ops.merge_samples(zz, znew, ss, snew), args.iters)
        add(f"K7 merge B={b} n=112+16", "gens_merge_samples", per, b * 128 * 16)
        # ---- round 4: one launch per sampling round (merge of round i + up-sampling of round i + 1; last round: merge + mid-points)
        for rnd, n in enumerate((64, 80, 96)):
            zz3, ss3 = z128[:, :n].contiguous(), sdf[:, :n].contiguous()
            _, vin3 = ops.ray_points(ro, rd, zz3, mset)
            za = (zz3[:, :16] + 1e-3).contiguous()
            sa, va = ss3[:, :16].contiguous(), vin3.reshape(b, n)[:, :16].contiguous()
            per = measure(lambda: ops.merge_upsample(ro, rd, zz3, ss3, vin3.reshape(b, n), za, sa, va, 16, mset, 128.0 * 2 ** rnd), args.iters)
            add(f"K7+K5/K6 merge_upsample B={b} n={n}+16->16", "gens_merge_upsample", per, b * (9 * (n + 16) + 9 * (n + 16) + 17 * 16 + 24),
                "merge of round i and up-sampling of round i + 1 in one launch")
        per = measure(lambda: ops.merge_mid_points(ro, rd, zz, znew, mset, 1 / 32), args.iters)
        add(f"K7+K3 merge_mid_points B={b} n=112+16", "gens_merge_mid_points", per, b * (4 * 128 + 17 * 128 + 24), "the last merge + render_core's mid-points and masks")

        # ---- K8 compositing forward (inference: no smooth vector)
        n = 128
        sd = (torch.rand(b, n, device=dev) - 0.3) * 0.2
        gr = torch.nn.functional.normalize(torch.randn(b, n, 3, device=dev), dim=-1)
        col = torch.rand(b, n, 3, device=dev)
        vm = valid.reshape(b, n)
        sv = torch.rand(b * n, s, device=dev) > 0.3
        inv_s = torch.tensor([200.0], device=dev)
        per = measure(lambda: ops.composite(ro, rd, z128, 1 / 32, sd, gr, None, col, vm, sv, inv_s, 1.0, c2ws[0]), args.iters)
        add(f"K8 composite fwd B={b} n=128 S=4", "gens_composite_fwd", per, b * n * (41 + s) + 100 * b)

    # ---- K8 backward (training, 512 rays) and at chunk size
    for bb, tag in ((512, "train step"), (b, "chunk")):
        sdg = sd[:bb].clone().requires_grad_(True)
        grg = gr[:bb].clone().requires_grad_(True)
        cog = col[:bb].clone().requires_grad_(True)
        smg = torch.randn(bb, n, 3, device=dev).requires_grad_(True)
        isg = inv_s.clone().requires_grad_(True)

        def comp_bwd():
            o = ops.composite(ro[:bb], rd[:bb], z128[:bb], 1 / 32, sdg, grg, smg, cog, vm[:bb], sv[:bb * n], isg, 1.0, c2ws[0])
            (o["color"].sum() + o["depth"].sum() + o["eik_num"].sum()).backward()
        per = measure(comp_bwd, max(20, args.iters // 2))
        add(f"K8 composite fwd (+smooth) B={bb} ({tag})", "gens_composite_fwd", per, bb * n * (53 + s) + 100 * bb)
        add(f"K8 composite bwd B={bb} ({tag})", "gens_composite_bwd", per, bb * n * (41 + s + 40) + 100 * bb)

    with torch.no_grad():
        # ---- K10 TV, K11 lattice
        for lvl, d in enumerate(dims):
            per = measure(lambda: ops._TVLevel.apply(vols[lvl], masks[lvl]), max(20, args.iters // 4))
            add(f"K10 tv fwd D={d}", "gens_tv_fwd", per, 20 * d ** 3)
        cnt = 64 ** 3 * 4
        per = measure(lambda: ops.lattice_points([-1, -1, -1], [1, 1, 1], 512, 0, cnt, dev), args.iters)
        add(f"K11 lattice points N={cnt}", "gens_lattice_points", per, cnt * 12)

    # ---- K13 LNCC patch statistic, one training step's worth (512 rays, 4 source views, 121 samples, 12 channels)
    from gens_amd.losses import compute_LNCC
    rg = torch.rand(1, 512, 121, 12, device=dev, requires_grad=True)
    sg = torch.rand(4, 512, 121, 12, device=dev, requires_grad=True)

    def lncc_step():
        compute_LNCC(rg, sg).sum().backward()
        rg.grad = sg.grad = None
    per = measure(lncc_step, args.iters)
    add("K13 lncc fwd B=512 S=4", "gens_lncc_fwd", per, 512 * 121 * 12 * 5 * 4)
    add("K13 lncc bwd B=512 S=4", "gens_lncc_bwd", per, 2 * 512 * 121 * 12 * 5 * 4)

    with torch.no_grad():
        # ---- K12 marching cubes on a 512^3 lattice (the reference's mesh resolution): sphere field, ~0.8 M vertices
        n = 512
        ax = torch.arange(n, device=dev, dtype=torch.float32) - (n - 1) / 2
        u = 180.3 - torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
        per = measure(lambda: ops.marching_cubes(u, 0.0), 20, warm=3)
        add("K12 mc classify 512^3", "gens_mc_classify", per, n ** 3 * 8)
        add("K12 mc emit 512^3", "gens_mc_emit", per, n ** 3 * 15)
        del u

    smi = ""
    try:
        if args.no_smi:
            raise RuntimeError("skipped (--no-smi)")
        smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=30).stdout
    except Exception as e:  # noqa: BLE001
        smi = f"rocm-smi unavailable: {e}"
    out = {"device": torch.cuda.get_device_name(0), "hbm_peak_GBs": HBM, "protocol": f"20 warm-up + {args.iters} timed launches (K1/TV/bwd: fewer, >= 20), HIP events on the launch stream",
           "rows": rows, "rocm_smi": smi.splitlines()[:40]}
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
