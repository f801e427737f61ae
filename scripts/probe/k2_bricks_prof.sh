# per-kernel durations of the stand-alone K2 backward with the brick scatter (rocprofv3 --kernel-trace --stats of k2_bwd_sorted_probe.py)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk2b && rocprofv3 --kernel-trace --stats -d /tmp/pk2b -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/probe/k2_bwd_sorted_probe.py > /tmp/pk2b.log 2>&1
f=$(find /tmp/pk2b -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("brick", "lookup_")):
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:9.1f} us  min {float(r['MinNs']) / 1e3:9.1f}  max {float(r['MaxNs']) / 1e3:9.1f}")
PY
