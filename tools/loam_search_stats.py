"""Search counters of the LOAM translation unit (instrumented build: the same -DGPSCAL_STATS applied to loam.hip):
slot 1 = lo_search_kernel, slot 2 = lm_point_kernel.  Wave-level counts per wave of the kernel."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpscalibration_amd import Context, synth, _lib
nseg, nsweeps = 6, 30
ctx = Context(0)
W = synth.lidar_world(0, length=600.0)
segs, stamps = [], []
for sgm in range(nseg):
    sw, st, _ = synth.drive(W, nsweeps, seed=100 + sgm, n_az=1800, start=(20.0 * sgm, 0.3 * (sgm % 8)))
    segs.append(sw); stamps.append(st)
L = _lib.load()
NSTAT, STAT_ITERS = 24, 64
buf = (ctypes.c_ulonglong * (NSTAT * STAT_ITERS))()
L.gpscal_debug_stats_loam.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.gpscal_debug_stats_loam(buf, NSTAT * STAT_ITERS)
os.environ["GPSCAL_LOAM_PIPELINE"] = "0"
ctx.loam_run(segs, stamps)
L.gpscal_debug_stats_loam(buf, NSTAT * STAT_ITERS)
for slot, name in ((1, "lo_search_kernel"), (2, "lm_point_kernel")):
    c = [int(buf[slot * NSTAT + k]) for k in range(NSTAT)]
    print(name, "waves", c[2], "mono", c[21], "ring grids", c[22], "| level passes (wave)", c[3], "(lane)", c[4], "| rows (wave)", c[5],
          "(lane)", c[6], "| groups of 4 (wave)", c[7], "(lane)", c[16], "| candidates (lane)", c[8], "| levels", c[11:16], flush=True)
c = [int(buf[1 * NSTAT + k]) for k in range(NSTAT)]
if c[10] > 0:
    n0 = max(c[10], 1)
    print("lo_search_kernel surf tiles (%d): wave 0 whole tile %.1f us, of which the split nearest search %.1f us; ring searches: own ring (wave 1) %.1f us, adjacent rings (waves 2 + 3, per wave) %.1f us" % (
        c[10], c[19] / n0 / 100.0, c[21] / n0 / 100.0, c[22] / n0 / 100.0, c[23] / n0 / 200.0))
