"""Timeline of the kernels of the LAST scan-batch build in a rocprofv3 --kernel-trace CSV (a build starts with
pack_points_kernel after the previous build's fill_warm_kernel): start, duration and the gap to the previous kernel."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last build: from the pack_points_kernel in front of the last self_nn_kernel (the target clouds) to the end
nn = max(i for i, r in enumerate(rows) if "self_nn" in r["Kernel_Name"])
first = max(i for i, r in enumerate(rows[:nn]) if "pack_points" in r["Kernel_Name"])
run = rows[first:]
t0 = int(run[0]["Start_Timestamp"])
prev = t0
busy = 0
for r in run:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0].replace("gpscal::", "")[:40]
    print("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((a - t0) / 1e3, (a - prev) / 1e3, (b - a) / 1e3, n))
    prev = max(prev, b)
    busy += b - a
print("span %.1f us, busy %.1f us" % ((prev - t0) / 1e3, busy / 1e3))
