"""Pooled per-call grid sets (GPSCAL_POOL_GRIDS=1): the poison probe's workload in child processes, twice
unpoisoned and under two fill bytes; every run must give identical results."""
import os, subprocess, sys
env = dict(os.environ, GPSCAL_POOL_GRIDS="1", PROBE_TAGS="p41,pff")
r = subprocess.run([sys.executable, "tools/poison_probe.py"], env=env)
print("pooled poison probe rc", r.returncode, flush=True)
# run-to-run: unpoisoned pooled vs unpoisoned hipMalloc
import pickle
import numpy as np
sys.path.insert(0, "tools")
import importlib.util
spec = importlib.util.spec_from_file_location("pp", "tools/poison_probe.py")
out = {}
for tag, pool in (("a", "1"), ("b", "1"), ("c", "0")):
    e = dict(os.environ, GPSCAL_POOL_GRIDS=pool)
    e.pop("GPSCAL_POISON", None)
    subprocess.run([sys.executable, "tools/poison_probe.py", "run_" + tag], env=e, check=True)
    out[tag] = pickle.load(open("/tmp/poisonp/run_%s.pkl" % tag, "rb"))
def same(x, y):
    if isinstance(x, dict):
        return x.keys() == y.keys() and all(same(x[k], y[k]) for k in x)
    if isinstance(x, (list, tuple)):
        return len(x) == len(y) and all(same(a, b) for a, b in zip(x, y))
    if isinstance(x, np.ndarray):
        return x.shape == y.shape and np.array_equal(x, y, equal_nan=True)
    return x == y
for k in out["a"]:
    print(k, "pooled run a == pooled run b:", same(out["a"][k], out["b"][k]), " pooled == hipMalloc:", same(out["a"][k], out["c"][k]), flush=True)
