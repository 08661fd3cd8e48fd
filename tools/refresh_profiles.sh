#!/bin/bash
# End-of-round refresh on the GPU box: PMC traffic of the step kernel for the current kernel sources, the full bench
# line (quotes that traffic), and the rocprofv3 summary of the bench command.  Outputs under gpurun_out/refresh/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
bash $R/tools/icp_traffic_iter.sh $O/pmc 50 || exit 1
cd $R
python tools/pmc_traffic_report.py $O/pmc 4194304 pairs64_points65536 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/icp_iter_run.py 50, end of round 2" > $O/pmc_traffic.json || exit 1
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-track --no-loam --no-single-pair > $O/bench_profiled.json 2> $O/bench_profiled.err || exit 1
cd $R
python tools/rocprof_summary.py $O/prof $O/bench_profiled.json > $O/bench_kernel_summary.json
cp $(ls $O/prof/*/*kernel_stats.csv $O/prof/*kernel_stats.csv 2>/dev/null | head -1) $O/bench_kernel_stats.csv
rm -rf $O/prof $O/pmc/*/*/*kernel_trace.csv
tail -c 600 $O/bench_full.json
