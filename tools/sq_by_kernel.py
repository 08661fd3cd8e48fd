"""Counters summed per kernel name over a run (tools/loam_sq.sh, tools/build_sq.sh): python tools/sq_by_kernel.py <dir>"""
import csv
import glob
import sys
from collections import OrderedDict, defaultdict
for pdir in sorted(glob.glob(sys.argv[1] + "/p*")):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for f in glob.glob(pdir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gpscal::", "")[:30]
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[n].add(r["Dispatch_Id"])
    names = sorted({c for v in acc.values() for c in v})
    print(pdir.split("/")[-1], "%-30s %6s " % ("kernel", "calls") + " ".join("%13s" % n[-13:] for n in names))
    key = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in names else names[0]
    for n in sorted(acc, key=lambda k: -acc[k].get(key, 0))[:14]:
        print("   %-30s %6d " % (n, len(calls[n])) + " ".join("%13.4g" % acc[n].get(c, 0) for c in names))
