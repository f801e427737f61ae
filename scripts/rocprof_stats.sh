#!/bin/bash
# rocprofv3 --kernel-trace --stats of one python workload; keeps only the per-kernel summary (the trace itself exceeds gpurun's 64 MiB).
# usage (GPU box, repository root): bash scripts/rocprof_stats.sh <tag> <script.py> [args...]   ->  gpurun_out/<tag>_kernel_stats.csv
set -e
ROOT=$(pwd)
TAG=$1
shift
SCRIPT=$ROOT/$1
shift
mkdir -p "$ROOT/gpurun_out"
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG --output-format csv -- python3 "$SCRIPT" "$@" > "$ROOT/gpurun_out/${TAG}_run.log" 2>&1 || true
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" \;
