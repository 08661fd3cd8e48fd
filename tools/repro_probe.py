"""Where does run-to-run variation of the ICP batch come from (system ROCm runtime: GPSCAL_NO_TORCH=1)?
(i) graph replays of one batch, (ii) eager (event-bracketed) runs of the same batch, (iii) separately built
batches, each compared bit for bit."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from gpscalibration_amd import Context, synth
ctx = Context(0)
print(ctx.info())
tg, to, sr, so, _ = synth.scan_batch(3, 20000)
def run(sb, iters, profile):
    sb.set_pose(None)
    T, err, _ = sb.icp(iters, profile=profile)
    idx, sqd = sb.correspondences()
    return T.copy(), err.copy(), idx.copy(), sqd.copy()
def same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))
sb = ctx.scan_batch(tg, to, sr, so)
for it in (1, 2, 12):
    g = [run(sb, it, False) for _ in range(6)]
    e = [run(sb, it, True) for _ in range(6)]
    print("iters %2d  graph replays equal: %s   eager runs equal: %s   graph == eager: %s" % (
        it, [same(g[0], x) for x in g[1:]], [same(e[0], x) for x in e[1:]], same(g[0], e[0])), flush=True)
builds = []
for _ in range(5):
    s2 = ctx.scan_batch(tg, to, sr, so)
    builds.append(run(s2, 12, True))
    s2.close()
print("separate builds (eager, 12 iterations) equal to the first:", [same(builds[0], x) for x in builds[1:]])
print("first build == original batch:", same(builds[0], run(sb, 12, True)))
sb.close(); ctx.close()
