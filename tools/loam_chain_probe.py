"""The bench's LOAM chain section alone (6 segments x 30 sweeps by default), for rocprofv3 kernel traces:
python tools/loam_chain_probe.py [nseg] [nsweeps] [repeats]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpscalibration_amd import Context, synth
nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nsweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = Context(0)
W = synth.lidar_world(0, length=max(600.0, 20.0 * nseg + 300.0))
segs, stamps = [], []
for sgm in range(nseg):
    sw, st, _ = synth.drive(W, nsweeps, seed=100 + sgm, n_az=1800, start=(20.0 * sgm, 0.3 * (sgm % 8)))
    segs.append(sw)
    stamps.append(st)
ctx.loam_run([segs[0][:4]], [stamps[0][:4]])
for r in range(reps):
    t0 = time.perf_counter()
    ctx.loam_run(segs, stamps)
    dt = time.perf_counter() - t0
    print("run %d: %d segments x %d sweeps in %.4f s = %.0f sweeps/s" % (r, nseg, nsweeps, dt, nseg * nsweeps / dt), flush=True)
