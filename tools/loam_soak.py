"""Soak: a long synthetic drive through gpscal_input_data_run against the CPU restatement
(not part of the test suite: the oracle needs ~1 minute)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gpscalibration_amd import Context, synth
import _oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
W = synth.lidar_world(2, length=0.9 * n + 100)
bag, st, truth = synth.drive(W, n, seed=9, n_az=900, speed=9.0, wiggle=0.05)
ctx = Context(0)
L, S, OV = 200.0, 80.0, 30.0
t0 = time.time(); got = ctx.input_data_run([bag], [st], L, S, OV); dg = time.time() - t0
t0 = time.time()
ref = [dict(t, flag=0) for t in O.input_data_pass(bag, st, L, 0.0)] + [dict(t, flag=1) for t in O.input_data_pass(bag, st, S, OV)]
dc = time.time() - t0
print("GPU %.2fs, CPU %.1fs" % (dg, dc))
print("gpu cuts", [(t["flag"], t["first"], t["last"]) for t in got])
print("cpu cuts", [(t["flag"], t["first"], t["last"]) for t in ref])
same = [(t["flag"], t["first"], t["last"]) for t in got] == [(t["flag"], t["first"], t["last"]) for t in ref]
print("same cuts:", same)
if same:
    print("max track diff %.3e m" % max(np.abs(a["track"][:, :2] - b["track"][:, :2]).max() for a, b in zip(got, ref)))
end = got[0]["track"][-1, :2] - got[0]["track"][0, :2]
print("first long track displacement", end, "truth", truth[got[0]["last"] - 1, :2] - truth[got[0]["first"], :2])
