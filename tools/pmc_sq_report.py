"""Per-launch SQ counters of the ICP step kernels from tools/icp_sq_iter.sh: python tools/pmc_sq_report.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict, OrderedDict

d = sys.argv[1]
rows = OrderedDict()
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "icp_step" not in r["Kernel_Name"]:
            continue
        key = (f.split("/p")[1][0], int(r["Dispatch_Id"]))
        rows.setdefault(key, {"k": "multi" if "multi" in r["Kernel_Name"] else "step", "grid": r["Grid_Size"]})
        rows[key][r["Counter_Name"]] = rows[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
by_pass = defaultdict(list)
for (p, did), v in rows.items():
    by_pass[p].append(v)
# the profiled run = the last launches of every pass; print them in order, one line per launch
for p in sorted(by_pass):
    ls = by_pass[p]
    names = [k for k in ls[0] if k not in ("k", "grid")]
    print("pass", p, "launches", len(ls))
    print("%4s %-5s %9s " % ("#", "kern", "grid") + " ".join("%14s" % n[-14:] for n in names))
    for i, v in enumerate(ls):
        print("%4d %-5s %9s " % (i, v["k"], v["grid"]) + " ".join("%14.4g" % v.get(n, 0) for n in names))
