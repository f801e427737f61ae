#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes) of the stand-alone gather kernels K2 forward (packed) and K4 forward.
# usage (GPU box, repository root): bash scripts/probe/gather_traffic.sh <out.json>
set -e
ROOT=$(pwd)
OUT=$1
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/pmc_gather
W="$ROOT/scripts/probe/gather_traffic_workload.py"
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_gather/fetch --output-format csv -- python3 $W > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_gather/write --output-format csv -- python3 $W > /dev/null 2>&1
cd "$ROOT"
python3 scripts/pmc_traffic.py /tmp/pmc_gather/fetch /tmp/pmc_gather/write "$OUT" "scripts/probe/gather_traffic_workload.py (K2 forward packed 4.19 M points x 3 levels; K4 forward 3.8 M points x 4 views x 5 levels)"
