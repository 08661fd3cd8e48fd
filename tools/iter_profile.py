"""Per-iteration launch times of icp_step_kernel (event-bracketed) for a few settings of GPSCAL_BALL_R."""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from gpscalibration_amd import Context, synth
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
balls = sys.argv[4].split(",") if len(sys.argv) > 4 else ["0", "2", "4"]
ctx = Context(0)
tg, to, sr, so, Tt = synth.scan_batch(npairs, n)
for r in balls:
    os.environ["GPSCAL_BALL_R"] = r
    sb = ctx.scan_batch(tg, to, sr, so)
    sb.icp(2); sb.set_pose(None)
    best = None
    for _ in range(3):
        sb.set_pose(None)
        _, _, ms = sb.icp(iters, profile=True)
        best = ms if best is None else np.minimum(best, ms)
    print("ball %s: %s | sum %.0f us mean %.1f" % (r, " ".join("%.0f" % (1e3 * v) for v in best), 1e3 * best.sum(), 1e3 * best.mean()), flush=True)
    sb.close()
ctx.close()
