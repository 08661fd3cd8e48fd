#!/bin/bash
# Shader-sequencer counters of the index-build kernels: tools/build_sq.sh <outdir> [pairs points]
out=$1; np=${2:-64}; pts=${3:-65536}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM"
n=1
for p in "$P1" "$P2"; do
  GPSCAL_BUILD_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $out/p$n -- python3 /root/repo/tools/build_probe.py $np $pts > $out.p$n.log 2>&1 || echo "pass $n failed"
  n=$((n+1))
done
