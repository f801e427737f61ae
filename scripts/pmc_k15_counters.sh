#!/bin/bash
# What bounds K15: SQ / TA / TCP / TCC counters of one 8 -> 8 block at 256^3 (scripts/probe/conv_pmc_workload.py), one rocprofv3 pass per group.
# usage (GPU box, repository root): bash scripts/pmc_k15_counters.sh > gpurun_out/k15_counters.txt
ROOT=$(pwd)
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/pmc_k15
i=0
for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD" \
             "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --pmc $group -d /tmp/pmc_k15/g$i --output-format csv -- python3 "$ROOT/scripts/probe/conv_pmc_workload.py" > /tmp/pmc_k15_$i.log 2>&1 || echo "group $i failed: $group"
done
cd "$ROOT"
python3 scripts/pmc_summary.py /tmp/pmc_k15 conv3d
