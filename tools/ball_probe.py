"""Compares the ball search (GPSCAL_BALL_R = 1..) with the fine -> coarse search on the benchmark batch:
results must be bit-identical, prints the per-iteration kernel times."""
import os
import sys
sys.path.insert(0, os.environ.get("AB_ROOT", "/root/repo"))
import numpy as np
import torch
from gpscalibration_amd import Context, synth

cases = [(64, 65536, 50), (16, 262144, 50), (4, 1048576, 50)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
rs = [int(v) for v in os.environ.get("PROBE_RS", "0,1,2,3").split(",")]
ctx = Context(0)
for npairs, n, iters in cases:
    tg, to, sr, so, _ = synth.scan_batch(npairs, n)
    dtg, dsr = torch.from_numpy(tg).cuda(), torch.from_numpy(sr).cuda()
    ref = None
    for R in rs:
        os.environ["GPSCAL_BALL_R"] = str(R)
        sb = ctx.scan_batch(dtg, to, dsr, so)
        sb.icp(iters, want_err=False)
        sb.set_pose(None)
        T, err, ms = sb.icp(iters, want_err=True, profile=True)
        idx, sqd = sb.correspondences()
        out = (T.copy(), err.copy(), idx, sqd)
        same = "ref"
        if ref is None:
            ref = out
        else:
            ok = all(np.array_equal(a, b) for a, b in zip(ref, out))
            same = "same" if ok else "DIFFERENT"
        print(f"{npairs}x{n} R={R}: sum {ms.sum()*1e3:9.0f} us mean {ms.mean()*1e3:7.1f} us  {same}  first: "
              + " ".join(f"{v*1e3:.0f}" for v in ms[:int(os.environ.get("PROBE_NSHOW", "12"))]), flush=True)
        sb.close()
