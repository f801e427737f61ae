#!/bin/bash
# HBM traffic (PMC) of K15 / K16 on one U-Net block at 256^3: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE) + scripts/pmc_traffic.py.
# usage (on the GPU box, from the repository root): bash scripts/pmc_unet_block.sh <out.json>
set -e
ROOT=$(pwd)
OUT=${1:-gpurun_out/pmc_traffic_unet_block.json}
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/pmc_unet
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_unet/fetch --output-format csv -- python3 "$ROOT/scripts/probe/conv_pmc_workload.py" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_unet/write --output-format csv -- python3 "$ROOT/scripts/probe/conv_pmc_workload.py" > /dev/null 2>&1
cd "$ROOT"
python3 scripts/pmc_traffic.py /tmp/pmc_unet/fetch /tmp/pmc_unet/write "$OUT" "scripts/probe/conv_pmc_workload.py (conv 8->8 stride 1 + instance-norm + ReLU at 256^3, forward + backward)"
