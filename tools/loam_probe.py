"""Times gpscal_loam_run_batched on synthetic drives and reports the distance to the CPU restatement."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gpscalibration_amd import Context, synth
import _oracle as O

nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nsw = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n_az = int(sys.argv[3]) if len(sys.argv) > 3 else 1800
W = synth.lidar_world(0, length=600.0)
segs, stamps = [], []
t0 = time.time()
for s in range(nseg):
    sw, st, truth = synth.drive(W, nsw, seed=100 + s, n_az=n_az, start=(20.0 * s, 0.3 * s))
    segs.append(sw); stamps.append(st)
print("generated %d x %d sweeps (%d pts each) in %.1fs" % (nseg, nsw, len(segs[0][0]), time.time() - t0), flush=True)
ctx = Context(0)
ctx.loam_run([segs[0][:4]], [stamps[0][:4]])  # warm-up
t0 = time.time(); got = ctx.loam_run(segs, stamps); dt = time.time() - t0
print("GPU: %.3fs for %d sweeps -> %.2f ms/sweep/segment-batch, %.1f sweeps/s" % (dt, nseg * nsw, 1e3 * dt / nsw, nseg * nsw / dt), flush=True)
t0 = time.time(); ref = O.loam_run(segs[0], stamps[0]); dc = time.time() - t0
print("CPU restatement: %.3fs for %d sweeps of one segment -> %.1f sweeps/s" % (dc, nsw, nsw / dc))
for key in ("lo_sum", "tm_mapped", "lm_aft"):
    m = np.isfinite(ref[key][:, 0])
    d = np.abs(got[0][key][m] - ref[key][m])
    print(key, "max rot diff %.2e  max trans diff %.2e" % (d[:, :3].max(), d[:, 3:].max()))
print("track diff %.2e" % np.abs(got[0]["track"][1:, :2] - ref["track"][1:, :2]).max())
print("final pose est", got[0]["tm_mapped"][-1], "truth", truth[-1])
