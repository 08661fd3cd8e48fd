#!/bin/bash
# Shader-sequencer counters of the ICP step kernels per launch: tools/icp_sq_iter.sh <outdir> <iters> [pairs points]
# (one --pmc pass per counter set: SQ issue / wait, LDS, SPI resource stalls, TA / cache)
out=$1; it=${2:-12}; np=${3:-64}; pts=${4:-65536}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
P3="SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_RES_STALL_CSN SPI_CSN_BUSY SPI_CSN_WAVE GRBM_GUI_ACTIVE GRBM_SPI_BUSY"
P4="TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
n=1
for p in "$P1" "$P2" "$P3" "$P4"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $out/p$n -- python3 /root/repo/tools/icp_iter_run.py $it $np $pts > $out.p$n.log 2>&1 || echo "pass $n failed"
  n=$((n+1))
done
