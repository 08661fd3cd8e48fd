"""k-NN on the collinear cloud that fails under the system ROCm runtime: how wrong, and does serialisation help?"""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import _oracle as O
from gpscalibration_amd import Context
ctx = Context(0)
print(ctx.info(), "NO_TORCH", os.environ.get("GPSCAL_NO_TORCH"), "SERIALIZE", os.environ.get("AMD_SERIALIZE_KERNEL"), "POISON", os.environ.get("GPSCAL_POISON"))
rng = np.random.default_rng(1)
line = np.c_[np.linspace(0, 100, 5000), np.zeros(5000), np.zeros(5000)].astype(np.float32)
qq = (rng.uniform(-10, 110, size=(2000, 3)) * np.array([1, 0.05, 0.05])).astype(np.float32)
ri, rd = O.knn_brute(line, qq, 3)
for trial in range(3):
    ix = ctx.knn_index(line)
    gi, gd = ix.search(qq, 3)
    bad = np.flatnonzero((gi != ri).any(axis=1))
    print("trial %d: wrong rows %d of %d; first wrong rows: %s" % (trial, len(bad), len(qq), bad[:8]))
    if len(bad):
        print("   got", gi[bad[:4]].tolist(), "want", ri[bad[:4]].tolist(), "got d", gd[bad[:2]].tolist(), "want d", rd[bad[:2]].tolist())
    ix.close()
ctx.close()
