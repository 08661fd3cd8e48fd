#!/bin/bash
# HBM bytes of icp_step_kernel per iteration (FETCH_SIZE / WRITE_SIZE in separate passes): tools/icp_traffic_iter.sh <outdir> <iters> [pairs points]
out=$1; it=${2:-30}; np=${3:-64}; pts=${4:-65536}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 /root/repo/tools/icp_iter_run.py $it $np $pts > $out.$c.log 2>&1 || echo "$c pass failed"
done
