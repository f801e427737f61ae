#!/usr/bin/env python3
"""Both roofs in one table: adds the vector-ALU figures of scripts/pmc_valu.py (per device kernel, aggregated over the same kernel_bench.py run under the
PMC counters) to the rows of a scripts/kernel_bench.py table (per entry point and case).  usage: merge_valu_roof.py <kernels_isolated.json> <pmc_valu.json>

valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES-equivalent: GRBM_GUI_ACTIVE / 8 XCDs x 1 024 SIMDs): the share of the kernel's duration in which
a SIMD's vector ALU was executing; `roof` names the one the kernel sits under: "valu" from 60 % busy, otherwise "hbm" from 40 % of the HBM peak by algorithmic
bytes, otherwise "latency" (neither unit is busy: launch size, dependent chains, atomics)."""
import json
import sys

DEVICE_KERNEL = {
    "gens_volume_build_fwd": "volume_build_fwd_lean_k", "gens_volume_build_bwd_levels": "volume_bwd_tiles_k", "gens_ray_points": "ray_points_k",
    "gens_lookup_volume_bwd": "lookup_bwd_k<0>", "gens_lookup_volume_bwd2": "lookup_bwd2_k<0>", "gens_lookup_feature_fwd": "lookup_feature_fwd_k<5>",
    "gens_upsample": "upsample_k", "gens_merge_samples": "merge_k", "gens_merge_upsample": "merge_upsample_k<0>", "gens_merge_mid_points": "merge_upsample_k<1>",
    "gens_composite_fwd": "composite_fwd_k", "gens_composite_bwd": "composite_bwd_k", "gens_tv_fwd": "tv_fwd4_k", "gens_lattice_points": "lattice_k",
    "gens_lncc_fwd": "lncc_fwd_k", "gens_lncc_bwd": "lncc_bwd_k", "gens_mc_classify": "mc_classify4_k", "gens_mc_emit": "mc_emit_k",
}


def main(table_path, valu_path):
    table, valu = json.load(open(table_path)), json.load(open(valu_path))["kernels"]
    for row in table["rows"]:
        name = DEVICE_KERNEL.get(row["kernel"])
        if row["kernel"] == "gens_lookup_volume_fwd":
            name = "lookup_fwd_paired_k" if "packed" in row["case"] else "lookup_fwd_k<0>"
        k = valu.get(name)
        if k is None:
            continue
        row["device_kernel"] = name
        row["valu_busy_frac"] = round(k["valu_busy_frac"], 3)
        row["valu_insts_per_wave"] = k["valu_insts_per_wave"]
        row["roof"] = "valu" if k["valu_busy_frac"] >= 0.6 else "hbm" if row["hbm_frac"] >= 0.4 else "latency"
    table["protocol"] = str(table.get("protocol", "")) + "  valu_busy_frac / valu_insts_per_wave / roof: scripts/merge_valu_roof.py from scripts/pmc_valu.py's counters (per device kernel, all cases of it together)."
    json.dump(table, open(table_path, "w"), indent=1)
    for row in table["rows"]:
        print("%-52s hbm %5.1f %%   valu %s   %s" % (row["case"][:52], 100 * row["hbm_frac"], ("%5.1f %%" % (100 * row["valu_busy_frac"])) if "valu_busy_frac" in row else "  —   ",
                                                     row.get("roof", "")))


if __name__ == "__main__":
    main(*sys.argv[1:3])
