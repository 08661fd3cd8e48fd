"""Counters of the kernels of the last index build from tools/build_sq.sh: python tools/build_sq_report.py <dir>"""
import csv
import glob
import sys
from collections import OrderedDict
for pdir in sorted(glob.glob(sys.argv[1] + "/p*")):
    rows = OrderedDict()
    for f in glob.glob(pdir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            rows.setdefault(k, {"name": r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gpscal::", "")[:28], "grid": r["Grid_Size"]})
            rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ks = sorted(rows)
    # the last build: from the last pack_points_kernel that precedes the last self_nn_kernel
    nn = max(i for i, k in enumerate(ks) if "self_nn" in rows[k]["name"])
    first = max(i for i, k in enumerate(ks[:nn]) if "pack_points" in rows[ks[i]]["name"])
    names = [n for n in rows[ks[first]] if n not in ("name", "grid")]
    print(pdir.split("/")[-1], " ".join("%13s" % n[-13:] for n in names))
    for k in ks[first:]:
        v = rows[k]
        if v["name"].startswith("__amd"):
            continue
        print("%-28s %9s " % (v["name"], v["grid"]) + " ".join("%13.4g" % v.get(n, 0) for n in names))
