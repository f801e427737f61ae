# durations of every K18 backward launch of a hot-path run, in launch order (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk18c && rocprofv3 --kernel-trace -d /tmp/pk18c -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/probe/k18_time.py > /dev/null 2>&1
f=$(find /tmp/pk18c -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "blend_train_k" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(" ".join(str((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) // 1000) for r in rows), "us")
PY
