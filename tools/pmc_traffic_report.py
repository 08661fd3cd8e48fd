#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of icp_step_kernel per launch and per iteration -> HBM bytes.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B fabric request, i.e.
exactly half the bytes of wide coalesced reads -> doubled here; WRITE_SIZE is exact.  Both counters are in KiB.
Usage: pmc_traffic_report.py <outdir with FETCH_SIZE/ and WRITE_SIZE/ passes> <queries per iteration> <key> <note> [iterations of the last run]"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha16  # noqa: E402

outdir, grid, key, note = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
last_iters = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # keep only the last run of that many iterations (the profiled one)
per = {}
total_queries = int(grid)  # queries of one iteration (= threads of a whole-batch launch of icp_step_kernel)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (outdir, c)):
        rows = [r for r in csv.DictReader(open(f))
                if ("icp_step_kernel" in r["Kernel_Name"] or "icp_step_multi_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == c]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        # an iteration = consecutive launches that together cover every query once (one whole-batch launch, or one launch
        # per chain while the chains run different kernels; the multi-query kernel covers two queries per thread)
        covered, acc, vals = 0, 0.0, []
        for r in rows:
            covered += int(r["Grid_Size"]) * (2 if "icp_step_multi_kernel" in r["Kernel_Name"] else 1)
            acc += float(r["Counter_Value"])
            if covered >= total_queries:
                vals.append(acc)
                covered, acc = 0, 0.0
    per[c] = vals[-last_iters:] if last_iters else vals
n = min(len(per["FETCH_SIZE"]), len(per["WRITE_SIZE"]))
by_it = [(2.0 * per["FETCH_SIZE"][i] + per["WRITE_SIZE"][i]) * 1024.0 for i in range(n)]
rec = {"FETCH_SIZE_KiB_avg": sum(per["FETCH_SIZE"][:n]) / n, "WRITE_SIZE_KiB_avg": sum(per["WRITE_SIZE"][:n]) / n,
       "launches": n, "hbm_bytes_per_launch": sum(by_it) / n, "hbm_bytes_by_iteration": by_it,
       "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE x1",
       "kernel_source_sha16": kernel_source_sha16(), "collected": note}
print(json.dumps({key: rec}, indent=1))
