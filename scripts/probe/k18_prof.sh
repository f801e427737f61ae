# per-kernel durations of a hot-path training step (rocprofv3 --kernel-trace --stats): K18 backward, its reduce, K14
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk18 && rocprofv3 --kernel-trace --stats -d /tmp/pk18 -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/probe/k18_time.py > /dev/null 2>&1
f=$(find /tmp/pk18 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("blend_train", "gemm_tn", "sdf_train_bwd")):
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:9.1f} us  min {float(r['MinNs']) / 1e3:9.1f}")
PY
