#!/bin/bash
# rocprofv3 kernel statistics of the bench with only the LOAM / bag -> KML sections heavy (2 pairs of 4096 points for the ICP part)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/bagprof -- python3 /root/repo/bench.py --steps 1 --warmup 1 --pairs 2 --points 4096 --iters 5 --no-cpu-baseline --no-single-pair > /root/repo/gpurun_out/bagprof.json 2> /root/repo/gpurun_out/bagprof.err
cd /root/repo
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/bagprof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-46s calls %6s avg %9.1f us total %8.2f ms %5.1f%%" % (r["Name"].split("(")[0].replace("void ", "").replace("gpscal::", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time %.1f ms" % (tot / 1e6))
PY
