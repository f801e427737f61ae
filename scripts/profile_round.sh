#!/bin/bash
# Everything the profiles/ directory holds for a round, in one GPU call (repository root):  bash scripts/profile_round.sh <tag>
#   gpurun_out/<tag>_bench.json                 python bench.py (the line the driver records)
#   gpurun_out/<tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_pmc_traffic[_train].json   FETCH_SIZE / WRITE_SIZE passes (scripts/pmc_bench.sh)
#   gpurun_out/<tag>_kernels_isolated.json      scripts/kernel_bench.py
#   gpurun_out/<tag>_{ft,hot}_*                 scripts/train_trace.sh (fine-tune / hot-path training step: kernel stats, launches per step)
TAG=${1:-r03}
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bench && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b --output-format csv -- python3 $ROOT/bench.py > $ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2> /dev/null; find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $ROOT/gpurun_out/${TAG}_bench_kernel_stats.csv \;)
timeout 900 bash scripts/pmc_bench.sh $TAG
timeout 600 python scripts/kernel_bench.py --out gpurun_out/${TAG}_kernels_isolated.json > gpurun_out/${TAG}_kernels_isolated.log 2>&1
timeout 300 bash scripts/train_trace.sh ${TAG}_ft --finetune
timeout 300 bash scripts/train_trace.sh ${TAG}_hot
ls -la gpurun_out | grep $TAG
