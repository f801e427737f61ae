#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE -d <out>/fetch --output-format csv -- python3 <repo>/bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-kernel-timing
    rocprofv3 --pmc WRITE_SIZE -d <out>/write --output-format csv -- python3 <repo>/bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-kernel-timing
    python scripts/pmc_traffic.py <out>/fetch <out>/write profiles/rNN_pmc_traffic.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: the counters are in KB (x 1024); on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads, so the corrected figure is 2 x FETCH_SIZE + WRITE_SIZE (an upper
estimate for gather-dominated kernels, whose 16-B accesses the guide calls uncalibrated).  Infinity-Cache hits are counted."""
import csv
import glob
import json
import sys
from collections import defaultdict

ENTRY = {"sdf_mlp_k": "gens_sdf_mlp", "sdf_value_t_k": "gens_sdf_value", "sdf_grad_t_k": "gens_sdf_grad", "sdf_value_h_k": "gens_sdf_value_f16", "sdf_grad_h_k": "gens_sdf_grad_f16", "blend_k": "gens_blend_views", "blend_t_k": "gens_blend_views", "composite_fwd_k": "gens_composite_fwd",
         "upsample_k": "gens_upsample", "merge_k": "gens_merge_samples", "volume_build_fwd_k": "gens_volume_build_fwd", "volume_build_fwd_lean_k": "gens_volume_build_fwd", "volume_build_fwd_levels_k": "gens_volume_build_levels", "volume_build_fwd_pow2_k": "gens_volume_build_fwd",
         "ray_points_k": "gens_ray_points", "compact_count_k": "gens_compact_valid", "compact_write_k": "gens_compact_valid",
         "compact_scan_k": "gens_compact_valid", "compact_points_count_k": "gens_compact_valid", "compact_points_write_k": "gens_compact_valid", "mc_classify_k": "gens_mc_classify", "mc_emit_k": "gens_mc_emit",
         "conv3d_gather_k": "gens_conv3d_gather", "conv3d_scatter2_k": "gens_conv3d_scatter2", "conv3d_wgrad_k": "gens_conv3d_wgrad",
         "instnorm_stats_k": "gens_instnorm_stats", "instnorm_relu_fwd_k": "gens_instnorm_relu_fwd",
         "instnorm_relu_bwd_stats_k": "gens_instnorm_relu_bwd_stats", "instnorm_relu_bwd_k": "gens_instnorm_relu_bwd",
         "sdf_train_fwd_k": "gens_sdf_train_fwd", "sdf_train_bwd_k": "gens_sdf_train_bwd", "sdf_train_scatter_k": "gens_sdf_train_scatter",
         "sdf_train_pack_k": "gens_sdf_train_pack", "blend_train_k": "gens_blend_train", "gemm_tn_batch2_partial_k": "gens_gemm_tn_batch",
         "gemm_tn_batch_partial_k": "gens_gemm_tn_batch", "volume_build_bwd_k": "gens_volume_build_bwd", "tv_fwd4_k": "gens_tv_fwd",
         "tv_bwd4_k": "gens_tv_bwd", "lookup_feature_bwd_k": "gens_lookup_feature_bwd", "mc_classify4_k": "gens_mc_classify",
         # round 3: the all-level K1 backward (four device kernels per entry-point launch) and the step-boundary kernels
         "volume_bwd_plan_k": "gens_volume_build_bwd_levels", "volume_bwd_scan_k": "gens_volume_build_bwd_levels", "volume_bwd_fill_k": "gens_volume_build_bwd_levels",
         "volume_bwd_tiles_k": "gens_volume_build_bwd_levels", "sdf_train_wgrad_k": "gens_sdf_train_wgrad", "sdf_train_norm_k": "gens_sdf_train_pack_wn",
         "blend_wgrad_k": "gens_blend_train_wgrad", "tv_levels_fwd_k": "gens_tv_levels_fwd", "tv_levels_bwd_k": "gens_tv_levels_bwd",
         "patch_warp_fwd_k": "gens_patch_warp_fwd", "patch_warp_bwd_k": "gens_patch_warp_bwd", "loss_fwd_k": "gens_loss_fwd", "loss_bwd_k": "gens_loss_bwd",
         "lncc_fwd_k": "gens_lncc_fwd", "lncc_bwd_k": "gens_lncc_bwd", "composite_bwd_k": "gens_composite_bwd",
         # round 4
         "merge_upsample_k": "gens_merge_upsample", "lookup_fwd_k": "gens_lookup_volume_fwd", "lookup_feature_fwd_k": "gens_lookup_feature_fwd",
         "lookup_bwd_k": "gens_lookup_volume_bwd", "lookup_bwd2_k": "gens_lookup_volume_bwd2", "lookup_scatter_k": "gens_lookup_volume_scatter",
         "conv3d_wgrad_mfma_k": "gens_conv3d_wgrad", "dw_tile_k": "gens_depthwise_conv2d", "dw_wgrad_tile_k": "gens_depthwise_conv2d", "dw_dgrad_k": "gens_depthwise_conv2d",
         "bn_stats_k": "gens_batchnorm2d", "bn_apply_k": "gens_batchnorm2d", "bn_bwd_stats_k": "gens_batchnorm2d", "bn_bwd_apply_k": "gens_batchnorm2d"}
# entry points made of several device kernels: the launches of this one are the entry point's
ONE_PER_ENTRY = {"gens_volume_build_bwd_levels": "volume_bwd_tiles_k"}


def kernel_source_hash():
    """sha1 over the HIP sources of the library: a traffic file is only quoted by bench.py for the tree it was measured on."""
    import hashlib
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gens_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def collect(root, counter):
    tot, disp = defaultdict(float), defaultdict(set)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0].strip()
            if name in ENTRY:
                tot[name] += float(r["Counter_Value"]) * 1024.0
                disp[name].add(r["Dispatch_Id"])
    return tot, disp


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-kernel-timing, ray chunk 32768"
    ft, fd = collect(fetch_dir, "FETCH_SIZE")
    wt, wd = collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(ft) | set(wt)):
        e = ENTRY[k]
        d = kernels.setdefault(e, {"fetch_bytes": 0.0, "write_bytes": 0.0, "launches_f": 0, "launches_w": 0, "device_kernels": []})
        d["fetch_bytes"] += ft.get(k, 0.0)
        d["write_bytes"] += wt.get(k, 0.0)
        d["device_kernels"].append(k)
        if e in ONE_PER_ENTRY and k != ONE_PER_ENTRY[e]:
            continue
        if not k.startswith("compact_") or k in ("compact_scan_k", "compact_points_write_k"):      # one entry-point launch = 3 (2) device kernels for the compaction
            d["launches_f"] += len(fd.get(k, ()))
            d["launches_w"] += len(wd.get(k, ()))
    res = {}
    for e, d in kernels.items():
        nf, nw = max(1, d["launches_f"]), max(1, d["launches_w"])
        f, w = d["fetch_bytes"] / nf, d["write_bytes"] / nw
        res[e] = {"fetch_bytes_per_launch_raw": int(f), "write_bytes_per_launch_raw": int(w),
                  "traffic_bytes_per_launch_corrected": int(2 * f + w), "launches": nf, "device_kernels": d["device_kernels"]}
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of " + workload + "; "
            "KB x 1024; launch-weighted mean over the kernel's template instances; corrected = 2 x FETCH + WRITE "
            "(MI355X_MICROARCH.md, HBM: gfx950 FETCH_SIZE reports half of wide coalesced reads; 16-B gathers uncalibrated, so the "
            "corrected figure is an upper estimate there); Infinity-Cache hits are counted")
    json.dump({"_note": note, "kernel_source_hash": kernel_source_hash(), "kernels": res}, open(out, "w"), indent=1)
    for e, r in res.items():
        print(f"{e:24s} fetch {r['fetch_bytes_per_launch_raw'] / 1e6:9.2f} MB  write {r['write_bytes_per_launch_raw'] / 1e6:9.2f} MB  corrected {r['traffic_bytes_per_launch_corrected'] / 1e6:9.2f} MB  x{r['launches']}")


if __name__ == "__main__":
    main()
