#!/usr/bin/env python3
"""The VALU roof of the per-ray / gather kernels (round-4 review, item 5b): HBM is the wrong roof for kernels that are bound by their instruction
stream, so they are priced against the vector ALU's ISSUE peak too.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d <out> --output-format csv -- python3 <repo>/scripts/kernel_bench.py --iters 10 --no-smi
    python3 scripts/pmc_valu.py <out> profiles/rNN_pmc_valu.json

Per device kernel, averaged over its dispatches:
    valu_insts            SQ_INSTS_VALU: wave-level vector instructions issued
    gui_cycles            GRBM_GUI_ACTIVE / 8: shader-clock cycles the GPU was busy with the dispatch (rocprofv3 reports the SUM over the eight XCDs:
                          merge_upsample_k's 979 k "cycles" are 408 us at 2.4 GHz for a 44 - 51 us kernel; the SQ counters are chip-wide sums and are not
                          divided -- SQ_WAVES comes out as the number of waves launched)
    valu_issue_frac       2 x valu_insts / (1024 SIMDs x gui_cycles): a wave64 VALU instruction occupies its SIMD-32 for 2 cycles
                          (MI355X_MICROARCH.md: v_fma_f32 2 cyc), 256 CUs x 4 SIMDs; transcendental / quarter-rate and 64-bit instructions occupy it
                          longer, so this is a LOWER bound of how busy the vector ALUs were
    valu_busy_frac        4 x SQ_ACTIVE_INST_VALU / (1024 x gui_cycles): the SQ's own count of quad-cycles with a VALU instruction executing"""
import csv
import glob
import json
import sys
from collections import defaultdict

N_SIMD = 256 * 4
N_XCD = 8


def main():
    root, out = sys.argv[1:3]
    rows = defaultdict(lambda: defaultdict(dict))           # kernel -> dispatch -> counter -> value
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0].strip()
            d = rows[name][r["Dispatch_Id"]]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    res = {}
    for name, disp in rows.items():
        n = len(disp)
        mean = lambda c: sum(d.get(c, 0.0) for d in disp.values()) / n  # noqa: E731
        gui, insts, act = mean("GRBM_GUI_ACTIVE") / N_XCD, mean("SQ_INSTS_VALU"), mean("SQ_ACTIVE_INST_VALU")
        if gui <= 0 or insts <= 0:
            continue
        res[name] = {"dispatches": n, "valu_insts": int(insts), "waves": int(mean("SQ_WAVES")), "gui_cycles": int(gui),
                     "valu_insts_per_wave": round(insts / max(1.0, mean("SQ_WAVES")), 1),
                     "valu_issue_frac": round(2.0 * insts / (N_SIMD * gui), 4), "valu_busy_frac": round(4.0 * act / (N_SIMD * gui), 4),
                     "sq_busy_cycles": int(mean("SQ_BUSY_CYCLES"))}
    note = __doc__.split("Per device kernel")[1]
    json.dump({"_note": "per device kernel" + note, "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["valu_issue_frac"]))}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["valu_issue_frac"])[:40]:
        print(f"{k[:60]:60s} x{v['dispatches']:<4d} insts/wave {v['valu_insts_per_wave']:8.1f}  issue {v['valu_issue_frac']:.3f}  busy {v['valu_busy_frac']:.3f}")


if __name__ == "__main__":
    main()
