"""Iterations/s of small batches (1, 2, 4, 8 pairs of 65 536 points, 50 iterations, replayed graph) with the step's
row walk flat or not and the three workgroup sizes: python tools/small_batch_probe.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gpscalibration_amd import Context, synth
ctx = Context(0)
for npairs in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8):
    tg, to, sr, so, _ = synth.scan_batch(npairs, 65536)
    d_tg, d_sr = torch.from_numpy(tg).cuda(), torch.from_numpy(sr).cuda()
    d_T = torch.empty((npairs, 4, 4), dtype=torch.float64, device="cuda")
    ref = None
    for flat in ("0", "1"):
        for blk in ("128", "256", "512"):
            os.environ["GPSCAL_STEP_FLAT"] = flat
            os.environ["GPSCAL_STEP_BLOCK"] = blk
            sb = ctx.scan_batch(d_tg, to, d_sr, so)
            del os.environ["GPSCAL_STEP_FLAT"], os.environ["GPSCAL_STEP_BLOCK"]
            for _ in range(2):
                sb.set_pose(None)
                sb.icp(50, T_out=d_T)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(10):
                sb.set_pose(None)
                sb.icp(50, T_out=d_T)
            ctx.sync()
            dt = (time.perf_counter() - t0) / 10
            idx, sqd = sb.correspondences()
            same = True if ref is None else bool(np.array_equal(idx, ref[0]) and np.array_equal(sqd, ref[1]))
            if ref is None:
                ref = (idx, sqd)
            print("%d pair(s) flat %s block %s: %.1f k iterations/s (%.3f ms per run) correspondences equal %s" % (npairs, flat, blk, npairs * 50 / dt / 1e3, dt * 1e3, same), flush=True)
            sb.close()
