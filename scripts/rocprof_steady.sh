#!/bin/bash
# rocprofv3 --kernel-trace of one python workload, summarised over its LAST <seconds> only: the first steps of a process carry MIOpen's solver
# search (hundreds of naive / candidate kernels of 20 - 200 ms each) and plan construction, which a whole-process --stats table mixes into the
# per-step averages.  Keeps the summary, not the trace (which exceeds gpurun's 64 MiB).
# usage (GPU box, repository root): bash scripts/rocprof_steady.sh <tag> <seconds> <script.py> [args...]  ->  gpurun_out/<tag>_steady_kernel_stats.csv
set -e
ROOT=$(pwd)
TAG=$1
WINDOW=$2
shift 2
SCRIPT=$ROOT/$1
shift
mkdir -p "$ROOT/gpurun_out"
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o $TAG --output-format csv -- python3 "$SCRIPT" "$@" > "$ROOT/gpurun_out/${TAG}_run.log" 2>&1 || true
TRACE=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$TRACE" "$WINDOW" "$ROOT/gpurun_out/${TAG}_steady_kernel_stats.csv" <<'PY'
import csv
import sys
from collections import defaultdict
trace, window, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(trace)))
end = max(int(r["End_Timestamp"]) for r in rows)
t0 = end - int(window * 1e9)
agg = defaultdict(lambda: [0, 0, 10 ** 18, 0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0:
        continue
    a = agg[r["Kernel_Name"]]
    d = e - s
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
total = sum(a[1] for a in agg.values())
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "WindowSeconds"])
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name, a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / total, 3), a[2], a[3], window])
print("kernels in the last %.2f s: %d launches, %.1f ms busy" % (window, sum(a[0] for a in agg.values()), total / 1e6))
# where the device waits for the host: idle time between consecutive kernels of the window, by the kernel that ENDS the gap
win = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if int(r["Start_Timestamp"]) >= t0))
gaps = defaultdict(lambda: [0, 0])
busy_end, idle = win[0][1], 0
for s, e, name in win[1:]:
    if s > busy_end:
        g = gaps[name]
        g[0] += 1
        g[1] += s - busy_end
        idle += s - busy_end
    busy_end = max(busy_end, e)
with open(out.replace("_kernel_stats.csv", "_gaps.txt"), "w") as f:
    f.write("idle %.1f ms of the last %.2f s (%.1f %%); by the kernel that follows the gap:\n" % (idle / 1e6, window, idle / (window * 1e7)))
    for name, g in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        f.write("%9.3f ms %6d gaps  %s\n" % (g[1] / 1e6, g[0], name[:110]))
PY
