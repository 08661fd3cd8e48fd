"""Runs the concurrent LOAM chain and the replay + segmentation call repeatedly and compares every output of every
repetition with the first one bit for bit (the two host threads and streams must not make the results depend on timing)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gpscalibration_amd import Context, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = Context(0)
W = synth.lidar_world(0, length=600.0)
segs, stamps = [], []
for sgm in range(5):
    sw, st, _ = synth.drive(W, 12 + 3 * sgm, seed=100 + sgm, n_az=900, start=(20.0 * sgm, 0.3 * sgm))
    segs.append(sw); stamps.append(st)
first = ctx.loam_run(segs, stamps)
bad = 0
for r in range(reps):
    got = ctx.loam_run(segs, stamps)
    for a, b in zip(first, got):
        for k in a:
            if not np.array_equal(a[k], b[k], equal_nan=True):
                bad += 1
print("loam_run: %d repetitions, %d differing arrays" % (reps, bad), flush=True)
bag_a, st_a, _ = synth.drive(W, 60, seed=1, n_az=900)
bag_b, st_b, _ = synth.drive(W, 35, seed=7, n_az=900, start=(200.0, -1.0), speed=6.0)
f0 = ctx.input_data_run([bag_a, bag_b], [st_a, st_b], 30.0, 14.0, 5.0, corner_pool_cap=1 << 16, surf_pool_cap=1 << 18)
bad2 = 0
for r in range(max(reps // 2, 1)):
    g = ctx.input_data_run([bag_a, bag_b], [st_a, st_b], 30.0, 14.0, 5.0, corner_pool_cap=1 << 16, surf_pool_cap=1 << 18)
    if len(g) != len(f0):
        bad2 += 1
        continue
    for a, b in zip(f0, g):
        if (a["flag"], a["bag"], a["first"], a["last"]) != (b["flag"], b["bag"], b["first"], b["last"]) or \
                not np.array_equal(a["track"], b["track"], equal_nan=True):
            bad2 += 1
print("input_data_run: %d repetitions, %d differing tracks" % (max(reps // 2, 1), bad2), flush=True)
sys.exit(1 if bad or bad2 else 0)
