import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gpscalibration_amd import Context, synth
ctx = Context(0)
npairs, n = 2, 65536
tg, to, sr, so, _ = synth.scan_batch(npairs, n)
dtg, dsr = torch.from_numpy(tg).cuda(), torch.from_numpy(sr).cuda()
res = {}
for iters in (1, 2):
    for tag, R in (("a", 0), ("b", 0), ("c", 2), ("d", 2), ("e", 1)):
        os.environ["GPSCAL_BALL_R"] = str(R)
        sb = ctx.scan_batch(dtg, to, dsr, so)
        T, err, _ = sb.icp(iters)
        idx, sqd = sb.correspondences()
        res[tag] = (T.copy(), err.copy(), idx.copy(), sqd.copy())
        sb.close()
    for x, y in (("a", "b"), ("c", "d"), ("a", "c"), ("a", "e")):
        a, b = res[x], res[y]
        print(f"iters {iters} {x} vs {y}: T maxdiff {np.abs(a[0]-b[0]).max():.3e} err maxdiff {np.abs(a[1]-b[1]).max():.3e} idx mism {(a[2]!=b[2]).sum()}", flush=True)
    print(res["a"][1], res["c"][1])
