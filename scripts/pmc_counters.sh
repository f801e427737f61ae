#!/bin/bash
# Issue / wait / cache counters of one kernel: one rocprofv3 pass per counter group over a workload script, summed per kernel name.
# usage (GPU box, repository root): bash scripts/pmc_counters.sh <kernel-name-filter> <workload.py> [args...]  > gpurun_out/<x>_counters.txt
ROOT=$(pwd)
FILTER=$1
shift
WORK="$ROOT/$1"
shift
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/pmc_ctr
i=0
for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout 200 rocprofv3 --pmc $group -d /tmp/pmc_ctr/g$i --output-format csv -- python3 "$WORK" "$@" > /tmp/pmc_ctr_$i.log 2>&1 || echo "group $i failed: $group"
done
cd "$ROOT"
python3 scripts/pmc_summary.py /tmp/pmc_ctr "$FILTER"
