#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output of tools/pmc_passes.sh for one kernel (substring match)."""
import collections
import csv
import glob
import sys

outdir, kern = sys.argv[1], sys.argv[2]
which = sys.argv[3] if len(sys.argv) > 3 else "last"
agg = {}
for f in sorted(glob.glob(outdir + "/p*/*/*counter_collection.csv")):
    d = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            d[int(r["Dispatch_Id"])]["_dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ids = sorted(d)
    if not ids:
        continue
    pick = ids[-1] if which == "last" else ids[int(which)]
    for k, v in d[pick].items():
        agg[k if k != "_dur_us" else "_dur_us(" + f.split("/")[-3] + ")"] = v
for k in sorted(agg):
    print("%-40s %18.1f" % (k, agg[k]))
