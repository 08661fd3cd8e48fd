#!/bin/bash
# Shader-sequencer counters of the LOAM chain's kernels at many segments: tools/loam_sq.sh <outdir> [segments sweeps]
out=$1; ns=${2:-48}; nw=${3:-12}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE"
n=1
for p in "$P1" "$P2"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $out/p$n -- python3 /root/repo/tools/loam_chain_probe.py $ns $nw 1 hbm > $out.p$n.log 2>&1 || echo "pass $n failed"
  n=$((n+1))
done
