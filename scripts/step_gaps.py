#!/usr/bin/env python3
"""Where the GPU idles inside ONE training step: python scripts/step_gaps.py <dir-with-*kernel_trace.csv> <marker-kernel> [steps] [min_gap_us]
A step = from one launch of <marker-kernel> (a kernel every step runs exactly once, e.g. scene_setup_k) to the next; the last [steps] steps of the
trace are averaged.  Prints busy / idle per step and every gap >= min_gap_us as (kernel before, kernel after): count per step, mean, share."""
import csv
import glob
import sys
from collections import defaultdict

root, marker = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
min_gap = float(sys.argv[4]) if len(sys.argv) > 4 else 15.0
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
assert len(marks) > steps + 1, (len(marks), "launches of the marker")
lo, hi = marks[-steps - 1], marks[-1]
sel = rows[lo:hi]
span = rows[hi][0] - rows[lo][0]
busy = sum(e - s for s, e, _ in sel)
print(f"{steps} steps: {span / steps / 1e3:.1f} us per step, kernels busy {busy / steps / 1e3:.1f} us, idle {(span - busy) / steps / 1e3:.1f} us, {len(sel) / steps:.0f} launches per step")
gaps = defaultdict(lambda: [0, 0])
small = 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows[lo:hi], rows[lo + 1:hi + 1]):
    g = s1 - e0
    if g >= min_gap * 1e3:
        a = gaps[(n0, n1)]
        a[0] += 1
        a[1] += g
    elif g > 0:
        small += g
print(f"gaps under {min_gap:.0f} us: {small / steps / 1e3:.1f} us per step")
for (n0, n1), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1]):
    print(f"  {t / steps / 1e3:8.1f} us per step  ({c / steps:.1f} x {t / c / 1e3:.1f} us)   after {n0}   before {n1}")
