#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training-step bench; keeps the per-kernel summary, the gap report and the launch count per step.
# usage (GPU box, repository root): bash scripts/train_trace.sh <tag> [train_step_bench flags...]  -> gpurun_out/<tag>_{kernel_stats.csv,gaps.txt,run.log}
set -e
ROOT=$(pwd)
TAG=$1
shift
mkdir -p "$ROOT/gpurun_out"
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
STEPS=40
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG --output-format csv -- python3 "$ROOT/scripts/train_step_bench.py" "$@" --steps $STEPS --warm 5 > "$ROOT/gpurun_out/${TAG}_run.log" 2>&1 || true
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" \;
python3 "$ROOT/scripts/gap_report.py" /tmp/prof_$TAG 30 > "$ROOT/gpurun_out/${TAG}_gaps.txt" 2>&1 || true
python3 "$ROOT/scripts/step_gaps.py" /tmp/prof_$TAG ${MARKER:-scene_setup_k} 20 10 > "$ROOT/gpurun_out/${TAG}_step_gaps.txt" 2>&1 || true
python3 - "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" $STEPS >> "$ROOT/gpurun_out/${TAG}_gaps.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) + 5
calls = sum(int(r["Calls"]) for r in rows)
ns = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"--- {calls / steps:.1f} launches per step (all {steps} steps incl. warm-up and set-up), {ns / steps / 1e6:.3f} ms of kernels per step")
for r in rows[:40]:
    print(f"{int(r['Calls']) / steps:7.1f} {int(r['TotalDurationNs']) / steps / 1e3:9.1f} us  {r['Name'][:110]}")
PY
