"""Per-queue summary of a rocprofv3 --kernel-trace CSV for the last LOAM chain run in it (runs start with
scan_registration_kernel): busy time, span, kernels by total time, and the gaps between kernels by size."""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sr = [i for i, r in enumerate(rows) if "scan_registration" in r["Kernel_Name"]]
run = rows[sr[-1]:]
t0 = int(run[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in run)
print("run span %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(run)))
byq = collections.defaultdict(list)
for r in run:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    a, b = int(rs[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rs)
    gaps = []
    prev = None
    for r in rs:
        if prev is not None:
            gaps.append((int(r["Start_Timestamp"]) - prev) / 1e3)
        prev = max(prev or 0, int(r["End_Timestamp"]))
    big = [g for g in gaps if g > 200]
    mid = [g for g in gaps if 20 < g <= 200]
    small = [g for g in gaps if 0 < g <= 20]
    print("queue %s: %d kernels, busy %.2f ms, span %.2f..%.2f ms; gaps >200us: %d = %.2f ms, 20..200us: %d = %.2f ms, <20us: %d = %.2f ms" % (
        q, len(rs), busy / 1e6, (a - t0) / 1e6, (b - t0) / 1e6, len(big), sum(big) / 1e3, len(mid), sum(mid) / 1e3, len(small), sum(small) / 1e3))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        n = r["Kernel_Name"].split("(")[0].replace("gpscal::", "")[:34]
        agg[n][0] += 1
        agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:10]:
        print("    %-34s %5d  %8.2f ms  %7.1f us" % (n, c, t / 1e6, t / c / 1e3))
