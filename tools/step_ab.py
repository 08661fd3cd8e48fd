"""A/B of the ICP step: the one-kernel step of rounds 1-2 (GPSCAL_ICP_LEGACY=1) against the two-kernel step
(icp_light_kernel + icp_heavy_kernel), same batch, same process.  Prints the event-bracketed time of every
iteration's step (all of its launches), the wall time of the replayed graph, and compares poses, error history and
the last iteration's correspondences between the variants.

    python tools/step_ab.py [pairs] [points] [iters] [variant ...]

A variant is a comma-separated list of NAME=VALUE environment settings applied while the batch is created
(e.g. "GPSCAL_ICP_LEGACY=1", "GPSCAL_BOX_CPQ=24,GPSCAL_BOX_MIN=4"); "default" = none."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from gpscalibration_amd import Context, synth  # noqa: E402

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
variants = sys.argv[4:] or ["GPSCAL_ICP_LEGACY=1", "default"]

import torch  # noqa: E402

ctx = Context(0)
tg, to, sr, so, Tt = synth.scan_batch(npairs, n)
d_tg, d_sr = torch.from_numpy(tg).cuda(), torch.from_numpy(sr).cuda()
ref = None
for v in variants:
    env = {} if v == "default" else dict(kv.split("=", 1) for kv in v.split(","))
    for k, val in env.items():
        os.environ[k] = val
    sb = ctx.scan_batch(d_tg, to, d_sr, so)
    for k in env:
        del os.environ[k]
    best = None
    for _ in range(3):
        sb.set_pose(None)
        T, err, ms = sb.icp(iters, profile=True)
        best = ms if best is None else np.minimum(best, ms)
    idx, sqd = sb.correspondences()
    # the replayed graph, as the bench times it
    for _ in range(2):
        sb.set_pose(None)
        sb.icp(iters, want_err=True)
    ctx.sync()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        sb.set_pose(None)
        Tg, errg, _ = sb.icp(iters, want_err=True)
    ctx.sync()
    wall = (time.perf_counter() - t0) / reps
    print("%-40s %s" % (v, " ".join("%.0f" % (1e3 * x) for x in best)))
    print("%-40s sum %.0f us  mean %.1f us  frac %.3f | graph %.3f ms -> %.0f k it/s | graph==eager %s"
          % ("", 1e3 * best.sum(), 1e3 * best.mean(), npairs * 32 * n / (best.mean() * 1e-3) / 8e12,
             1e3 * wall, npairs * iters / wall / 1e3, bool(np.array_equal(T, Tg))), flush=True)
    if ref is None:
        ref = (T, err, idx, sqd)
    else:
        print("%-40s vs first variant: max|dT| %.2e  max|derr| %.2e  idx equal %s  sqd equal %s"
              % ("", np.abs(T - ref[0]).max(), np.abs(err - ref[1]).max(), bool(np.array_equal(idx, ref[2])),
                 bool(np.array_equal(sqd, ref[3]))), flush=True)
    sb.close()
ctx.close()
