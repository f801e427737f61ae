# per-kernel durations of K18t's backward and its reduction (rocprofv3 --kernel-trace --stats) at the training step's shape
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk18t && rocprofv3 --kernel-trace --stats -d /tmp/pk18t -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/probe/k18t_probe.py "$@" > /tmp/pk18t.log 2>&1
tail -3 /tmp/pk18t.log
f=$(find /tmp/pk18t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("blend_train", "reduce")):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:9.1f} us  min {float(r['MinNs']) / 1e3:9.1f}")
PY
