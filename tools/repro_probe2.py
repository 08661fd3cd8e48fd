"""Separately built ICP batches under the system ROCm runtime: are one-iteration results exact (vs the kd-tree
oracle) and bit-identical between builds?  Which array differs first?"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import _oracle as O
from gpscalibration_amd import Context, synth
ctx = Context(0)
tg, to, sr, so, _ = synth.scan_batch(3, 20000)
ref = [O.KdTree(tg[to[p]:to[p + 1]]).icp_iterate(sr[so[p]:so[p + 1]], np.eye(4)) for p in range(3)]
res = []
for b in range(6):
    sb = ctx.scan_batch(tg, to, sr, so)
    T, err, _ = sb.icp(1, profile=True)
    idx, sqd = sb.correspondences()
    T2, err2, _ = sb.icp(1, profile=True)   # second iteration (warm-started)
    idx2, sqd2 = sb.correspondences()
    res.append((T.copy(), err.copy(), idx, sqd, T2.copy(), err2.copy(), idx2, sqd2))
    ok = all(np.array_equal(idx[so[p]:so[p + 1]], ref[p][2]) and np.array_equal(sqd[so[p]:so[p + 1]], ref[p][3]) for p in range(3))
    dT = max(np.abs(T[p] - ref[p][0]).max() for p in range(3))
    print("build %d: iteration-1 correspondences exact vs oracle: %s, |T - oracle| %.2e" % (b, ok, dT), flush=True)
    sb.close()
names = ["T1", "err1", "idx1", "sqd1", "T2", "err2", "idx2", "sqd2"]
for b in range(1, 6):
    print("build %d vs 0:" % b, {n: bool(np.array_equal(x, y)) for n, x, y in zip(names, res[0], res[b])},
          "max|dT1| %.2e" % np.abs(res[0][0] - res[b][0]).max(), flush=True)
ctx.close()
