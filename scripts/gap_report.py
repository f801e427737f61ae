#!/usr/bin/env python3
"""GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace run: python scripts/gap_report.py <dir-with-*kernel_trace.csv> [top]
Prints the busy / idle split of the last third of the trace (steady state) and the largest gaps with the kernels on either side."""
import csv
import glob
import sys

root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
rows = rows[len(rows) * 2 // 3:]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(span - busy) / 1e6:.2f} ms")
gaps = []
hist = {}
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g > 0:
        gaps.append((g, n0, n1))
        key = (n0[:40], n1[:40])
        hist[key] = hist.get(key, 0) + g
for g, a, b in sorted(gaps, reverse=True)[:top]:
    print(f"{g / 1e3:9.1f} us   after {a}   before {b}")
print("--- idle by kernel pair")
for (a, b), g in sorted(hist.items(), key=lambda kv: -kv[1])[:top]:
    print(f"{g / 1e3:9.1f} us   {a} -> {b}")
