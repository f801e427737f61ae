#!/bin/bash
# usage (GPU box, repository root): bash scripts/probe/k1_bwd_counters.sh <tag>
# -> gpurun_out/<tag>/{counters.txt (issue / LDS / memory-side counters of the K1 backward kernels, 256^3 alone), probe.log, probe_conf.log, trace.txt}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_ctr /tmp/tr_k1
i=0
for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout 120 rocprofv3 --pmc $group -d /tmp/pmc_ctr/g$i --output-format csv -- python3 $ROOT/scripts/probe/k1_bwd_levels_probe.py --counters > /tmp/pmc_ctr_$i.log 2>&1 || echo "group $i failed"
done
timeout 180 rocprofv3 --kernel-trace -d /tmp/tr_k1 -o t --output-format csv -- python3 $ROOT/scripts/probe/k1_bwd_levels_probe.py > $OUT/probe.log 2>&1
cd $ROOT
python3 scripts/pmc_summary.py /tmp/pmc_ctr volume_bwd > $OUT/counters.txt 2>&1
python3 scripts/probe/k1_bwd_trace_summary.py $(find /tmp/tr_k1 -name '*kernel_trace.csv') > $OUT/trace.txt 2>&1
timeout 180 python3 scripts/probe/k1_bwd_levels_probe.py --conf-shape > $OUT/probe_conf.log 2>&1
grep -v amdgpu.ids $OUT/probe.log | tail -8; grep plan_k $OUT/trace.txt; tail -8 $OUT/probe_conf.log
