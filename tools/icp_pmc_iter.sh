#!/bin/bash
# per-iteration PMC of icp_step_kernel: tools/icp_pmc_iter.sh <outdir>
out=$1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 /root/repo/tools/icp_iter_run.py 12 > $out.p$i.log 2>&1 || echo "pass $i failed"
done
