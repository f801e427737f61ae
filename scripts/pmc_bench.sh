#!/bin/bash
# HBM traffic (PMC) of the bench command and of a training step: two rocprofv3 passes each (FETCH_SIZE, WRITE_SIZE cannot share a pass on
# gfx950) + scripts/pmc_traffic.py.   usage (GPU box, repository root): bash scripts/pmc_bench.sh <tag>   ->  gpurun_out/<tag>_pmc_traffic*.json
set -e
ROOT=$(pwd)
TAG=${1:-r02}
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
BENCH="$ROOT/bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-kernel-timing --headline-only"
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_$TAG/fetch --output-format csv -- python3 $BENCH > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_$TAG/write --output-format csv -- python3 $BENCH > /dev/null 2>&1
TRAIN="$ROOT/scripts/train_step_bench.py --steps 2 --warm 1 --no-auto"
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_$TAG/tfetch --output-format csv -- python3 $TRAIN > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_$TAG/twrite --output-format csv -- python3 $TRAIN > /dev/null 2>&1
cd "$ROOT"
python3 scripts/pmc_traffic.py /tmp/pmc_$TAG/fetch /tmp/pmc_$TAG/write gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.txt
python3 scripts/pmc_traffic.py /tmp/pmc_$TAG/tfetch /tmp/pmc_$TAG/twrite gpurun_out/${TAG}_pmc_traffic_train.json "scripts/train_step_bench.py --steps 2 --warm 1 (hot-path training step, 512 rays)" > gpurun_out/${TAG}_pmc_traffic_train.txt
