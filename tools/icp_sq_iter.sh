#!/bin/bash
# Shader-sequencer counters of the ICP step kernels per launch: tools/icp_sq_iter.sh <outdir> <iters> [pairs points]
# (two --pmc passes of 8 SQ counters; GPSCAL_ICP_MULTI_BELOW=0 keeps icp_step_kernel in every iteration)
out=$1; it=${2:-12}; np=${3:-64}; pts=${4:-65536}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P3="TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
n=1
for p in "$P1" "$P2" "$P3"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $out/p$n -- python3 /root/repo/tools/icp_iter_run.py $it $np $pts > $out.p$n.log 2>&1 || echo "pass $n failed"
  n=$((n+1))
done
