"""Mean launch time of icp_step_kernel over a run for a few index geometries (points per level-0 cell, level ratio,
ball radius) at one batch shape: python tools/cfg_sweep.py npairs points iters"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gpscalibration_amd import Context, synth
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tg, to, sr, so, _ = synth.scan_batch(npairs, n)
d_tg, d_sr = torch.from_numpy(tg).cuda(), torch.from_numpy(sr).cuda()
ctx = Context(0)
for ratio in ("2.5", "2.0", "1.7"):
    for per_cell in (3.0, 6.0, 12.0, 1.5):
        for ball in ("0", "3"):
            os.environ["GPSCAL_LEVEL_RATIO"] = ratio
            os.environ["GPSCAL_BALL_R"] = ball
            sb = ctx.scan_batch(d_tg, to, d_sr, so, cell_size=-per_cell)
            sb.icp(2)
            best = None
            for _ in range(2):
                sb.set_pose(None)
                _, _, ms = sb.icp(iters, profile=True)
                best = ms if best is None else np.minimum(best, ms)
            us = 1e3 * best
            print("ratio %s  %4.1f pts/cell  ball %s: mean %.1f us (frac %.3f)  first %.0f  it2-8 %.0f  last %.0f | build %.1f ms" % (
                ratio, per_cell, ball, us.mean(), npairs * 32.0 * n / (us.mean() * 1e-6) / 8e12, us[0], us[1:8].mean(), us[-1],
                1e3 * sb.build_seconds), flush=True)
            sb.close()
