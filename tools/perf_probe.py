#!/usr/bin/env python3
"""Per-iteration duration of the ICP correspondence kernel (event-bracketed) for a batch
of synthetic scan pairs; used while tuning.  `python tools/perf_probe.py --pairs 64`."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--points", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cell", type=float, default=0.0)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch

    from gpscalibration_amd import Context, synth
    ctx = Context(0)
    tg, to, sr, so, _ = synth.scan_batch(a.pairs, a.points)
    d_tg, d_sr = torch.from_numpy(tg).cuda(), torch.from_numpy(sr).cuda()
    sb = ctx.scan_batch(d_tg, to, d_sr, so, cell_size=a.cell)
    print("build %.4f s" % sb.build_seconds)
    best = None
    for _ in range(a.reps):
        sb.set_pose(None)
        T, err, ms = sb.icp(a.iters, profile=True)
        best = ms if best is None else np.minimum(best, ms)
    ab = a.pairs * 32 * a.points
    print("per-iteration ms:", " ".join("%.3f" % v for v in best))
    print("mean %.4f ms  -> %.1f GB/s algorithmic (%.2f%% of 8 TB/s); converged-iteration %.4f ms -> %.1f GB/s"
          % (best.mean(), ab / best.mean() / 1e6, ab / best.mean() / 1e6 / 80.0, best[-10:].mean(),
             ab / best[-10:].mean() / 1e6))
    print("mean err first/last: %.4f %.4f" % (err[:, 0].mean(), err[:, -1].mean()))
    sb.close()
    ctx.close()


if __name__ == "__main__":
    main()
