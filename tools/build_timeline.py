"""Kernel timeline of the last index build in a rocprofv3 kernel trace of tools/build_probe.py (start, gap to the
previous kernel's end -- negative where two streams overlap --, duration): python tools/build_timeline.py <trace dir or *_kernel_trace.csv>"""
import csv
import glob
import sys
rows = []
import os
files = [sys.argv[1]] if os.path.isfile(sys.argv[1]) else glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gpscal::", "")[:44]))
rows.sort()
# the last build = the kernels behind the last gap of more than 10 ms
start = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 10_000_000:
        start = i
sel = rows[start:]
t0 = sel[0][0]
prev_end = t0
busy = 0
for s, e, n in sel:
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n))
    busy += e - s
    prev_end = max(prev_end, e)
print("span %.1f us, kernels %.1f us, %d launches" % ((prev_end - t0) / 1e3, busy / 1e3, len(sel)))
