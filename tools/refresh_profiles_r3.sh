#!/bin/bash
# Round-3 refresh on the GPU box: PMC traffic of the step kernel at the three BASELINE sizes (separate FETCH_SIZE /
# WRITE_SIZE passes, no other trace domain), the full bench line, the rocprofv3 summary of the bench command, and the
# bench lines of configs[3]'s per-GPU share and configs[4]'s scan size.  Outputs under gpurun_out/refresh/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd $R
echo "{}" > $O/pmc_traffic.json
for cfg in "64 65536 50 4194304" "125 262144 20 32768000" "32 1048576 50 33554432"; do
  set -- $cfg
  bash $R/tools/icp_traffic_iter.sh $O/pmc_$1x$2 $3 $1 $2 || exit 1
  python tools/pmc_traffic_report.py $O/pmc_$1x$2 $4 pairs$1_points$2 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/icp_iter_run.py $3 $1 $2 (second run of the batch), round 3" $3 > $O/one.json || exit 1
  python - $O/pmc_traffic.json $O/one.json <<'PY'
import json, sys
a = json.load(open(sys.argv[1])); a.update(json.load(open(sys.argv[2]))); json.dump(a, open(sys.argv[1], "w"), indent=1)
PY
  rm -rf $O/pmc_$1x$2
  echo "pmc $1 x $2 done"
done
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err || exit 1
echo "bench done"
python bench.py --steps 5 --warmup 2 --points 262144 --pairs 125 --iters 20 --no-cpu-baseline --no-track --no-loam --no-single-pair > $O/bench_cfg3.json 2> $O/bench_cfg3.err || exit 1
python bench.py --steps 5 --warmup 2 --points 1048576 --pairs 32 --iters 50 --no-cpu-baseline --no-track --no-loam --no-single-pair > $O/bench_cfg4.json 2> $O/bench_cfg4.err || exit 1
echo "cfg3 / cfg4 done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-track --no-loam --no-single-pair > $O/bench_profiled.json 2> $O/bench_profiled.err || exit 1
cd $R
python tools/rocprof_summary.py $O/prof $O/bench_profiled.json > $O/bench_kernel_summary.json
cp $(ls $O/prof/*/*kernel_stats.csv $O/prof/*kernel_stats.csv 2>/dev/null | head -1) $O/bench_kernel_stats.csv
rm -rf $O/prof
tail -c 400 $O/bench_full.json
