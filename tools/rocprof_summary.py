#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace of bench.py: icp_step_kernel launches by grid size (whole-batch launches of
the roofline measurement, per-chain launches of the timed region), the solve kernel, and the agreement with the
bench's own HIP-event numbers.  Usage: rocprof_summary.py <trace dir> <bench json>"""
import csv
import glob
import json
import sys

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[-1]
bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
rows = list(csv.DictReader(open(f)))
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {"bench_roofline": bench["roofline"], "bench_value": bench["value"], "bench_ms_per_step": bench["ms_per_step"]}
groups = {}
STEP = ("icp_step_kernel", "icp_step_multi_kernel")  # the iterations' step kernels (the second from the learnt iteration on)
steps = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    if name in STEP + ("icp_solve_kernel",):
        g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
        groups.setdefault((name, g), []).append(d(r))
        if name in STEP:
            # queries a launch covers: one per thread, two per thread in the multi-query kernel
            steps.append((int(r["Start_Timestamp"]), d(r), g * (2 if name == "icp_step_multi_kernel" else 1)))
out["kernels"] = [{"kernel": k[0], "grid_threads": k[1], "launches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v),
                   "max_us": max(v), "vgpr": None} for k, v in sorted(groups.items())]
# The roofline is quoted for the profiling run, the LAST run of the command: its launches (whole batch, or one per chain
# in the few iterations in which the chains run different kernels) are the trace's last ones that together cover
# iterations x queries.
iters, queries = bench["config"]["iters"], bench["config"]["pairs_per_gpu"] * bench["config"]["points"]
steps.sort()
covered, total_us, n = 0, 0.0, 0
for _, dur, q in reversed(steps):
    if covered >= iters * queries:
        break
    covered += q
    total_us += dur
    n += 1
if covered >= iters * queries:
    out["profile_run_step_launches"] = n
    out["whole_batch_avg_us_rocprof"] = total_us / iters
    out["whole_batch_avg_us_bench_events"] = 1e3 * bench["roofline"]["avg_launch_ms"]
    out["roofline_frac_from_rocprof"] = bench["roofline"]["algorithmic_bytes_per_launch"] / (total_us / iters * 1e-6) / 8e12
print(json.dumps(out, indent=1))
