#!/bin/bash
# usage (GPU box, repository root): bash scripts/probe/k6gh_counters.sh <out.txt>
# Issue / wait / cache counters and HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the SDF-network kernels, float32 and split-half,
# value + gradient and value only, over scripts/probe/k6gh_probe.py --value (3.4 M points, volumes 256 / 128 / 64).
ROOT=$(pwd)
OUT=${1:-gpurun_out/k6gh_counters.txt}
mkdir -p $(dirname $OUT)
bash scripts/pmc_counters.sh sdf_ scripts/probe/k6gh_probe.py --reps 2 --value > $OUT 2>&1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_k6gh
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c -d /tmp/pmc_k6gh/$c --output-format csv -- python3 $ROOT/scripts/probe/k6gh_probe.py --reps 2 --value > /tmp/pmc_k6gh_$c.log 2>&1 || echo "$c pass failed" >> $ROOT/$OUT
done
cd $ROOT
echo "---- HBM traffic per launch (KB x 1024; corrected = 2 x FETCH + WRITE, MI355X_MICROARCH.md)" >> $OUT
python3 scripts/pmc_traffic.py /tmp/pmc_k6gh/FETCH_SIZE /tmp/pmc_k6gh/WRITE_SIZE /tmp/pmc_k6gh_traffic.json "scripts/probe/k6gh_probe.py --reps 2 --value (3.4 M points per launch)" >> $OUT 2>&1
