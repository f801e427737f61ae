#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name: python scripts/pmc_summary.py <dir> [name-filter]"""
import csv
import glob
import sys
from collections import defaultdict

root, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        if filt and filt not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k in acc:
    print(f"{k}  dispatches={len(cnt[k])}")
    for c, v in sorted(acc[k].items()):
        print(f"    {c:32s} {v / len(cnt[k]):16.0f} per dispatch")
