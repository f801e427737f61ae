#!/usr/bin/env python3
"""Why a gather / per-ray kernel sits below the 40 % HBM line: reads the per-kernel counter sums that scripts/pmc_counters.sh printed for
scripts/kernel_bench.py and adds a one-line citation ("why") to every row of an isolated-kernel table whose hbm_frac is below 0.40.

    bash scripts/pmc_counters.sh "" scripts/kernel_bench.py --iters 20 --no-smi > gpurun_out/kernels_counters.txt
    python scripts/kernel_counters.py gpurun_out/kernels_counters.txt profiles/rNN_kernels_isolated.json

Quantities (per device kernel, all its launches in the run): VALU issue = 4 x SQ_ACTIVE_INST_VALU / (SIMDs x kernel cycles) -- the share of
SIMD cycles that issued a vector instruction; wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES -- the share of wave time spent waiting for the
previous instruction's result or a memory return; VALU per wave and memory instructions per wave."""
import json
import re
import sys

ENTRY = {"gens_volume_build_fwd": "volume_build_fwd", "gens_ray_points": "ray_points_k", "gens_lookup_volume_fwd": "lookup_fwd_k",
         "gens_lookup_volume_bwd": "lookup_bwd_k", "gens_lookup_volume_bwd2": "lookup_bwd2_k", "gens_lookup_feature_fwd": "lookup_feature_fwd_k",
         "gens_upsample": "upsample_k", "gens_merge_samples": "merge_k", "gens_composite_fwd": "composite_fwd_k", "gens_composite_bwd": "composite_bwd_k",
         "gens_tv_fwd": "tv_fwd4_k", "gens_lattice_points": "lattice_points_k", "gens_lncc_fwd": "lncc_fwd_k", "gens_lncc_bwd": "lncc_bwd_k",
         "gens_mc_classify": "mc_classify4_k", "gens_mc_emit": "mc_emit_k"}


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+dispatches=(\d+)", line)
        if m:
            cur = out.setdefault(m.group(1).replace("void ", "").split("<")[0].strip(), {"dispatches": int(m.group(2))})
            continue
        m = re.match(r"^\s+(\S+)\s+(\d+) per dispatch", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return out


def main():
    counters = parse(sys.argv[1])
    table = json.load(open(sys.argv[2]))
    rows = table if isinstance(table, list) else table["rows"]
    for r in rows:
        if r.get("hbm_frac", 1.0) >= 0.40:
            continue
        key = ENTRY.get(r["kernel"])
        c = next((v for k, v in counters.items() if key and k.startswith(key)), None)
        if not c or "SQ_WAVE_CYCLES" not in c:
            r["why"] = "launch-latency bound at this size" if r["median_us"] < 12 else "no counters collected"
            continue
        cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                      # per XCD
        valu_issue = 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024.0 * cycles) if cycles else 0.0
        wait = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        waves = max(c.get("SQ_WAVES", 1.0), 1.0)
        r["why"] = (f"VALU issue {100 * valu_issue:.0f} % of SIMD cycles, waves wait {100 * wait:.0f} % of their time; "
                    f"{c.get('SQ_INSTS_VALU', 0.0) / waves:.0f} VALU + {c.get('SQ_INSTS_VMEM_RD', 0.0) / waves:.0f} memory-read instructions per wave "
                    f"(counters of the kernel's launches in scripts/kernel_bench.py; the mix of shapes where a kernel is listed at several)")
    json.dump(table, open(sys.argv[2], "w"), indent=1)
    for r in rows:
        if "why" in r:
            print(f"{r['case'][:44]:44s} {r['hbm_frac']:.3f}  {r['why'][:150]}")


if __name__ == "__main__":
    main()
