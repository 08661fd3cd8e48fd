#!/bin/bash
# instruction-cache counters of the LOAM kernels at many segments: tools/icache_probe.sh <outdir> <nseg>
out=$1; nseg=${2:-32}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_INST_LEVEL_[A-Z_]*" | sort -u > $out.avail.txt
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out -- python3 /root/repo/tools/loam_probe.py $nseg 10 1800 > $out.log 2>&1 || echo "pmc pass failed"
