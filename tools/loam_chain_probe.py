"""The bench's LOAM chain section alone (6 segments x 30 sweeps by default), for rocprofv3 kernel traces:
python tools/loam_chain_probe.py [nseg] [nsweeps] [repeats] [host|hbm|both]"""
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
import torch
ctx.loam_run([segs[0][:4]], [stamps[0][:4]])
t0 = time.perf_counter()
packed = ctx.loam_pack(segs, stamps)
print("packing the argument arrays (numpy): %.4f s, %.0f MB of sweeps" % (time.perf_counter() - t0, packed[0].nbytes / 1e6), flush=True)
mode = sys.argv[4] if len(sys.argv) > 4 else "both"  # host | hbm | both
resident = None
if mode != "host":
    resident = (torch.from_numpy(packed[0]).cuda(),) + packed[1:]
    torch.cuda.synchronize()
for r in range(reps):
    msg = "run %d: %d segments x %d sweeps" % (r, nseg, nsweeps)
    if mode != "hbm":
        t0 = time.perf_counter()
        ctx.loam_run_packed(packed)
        dt = time.perf_counter() - t0
        msg += " in %.4f s = %.0f sweeps/s (sweeps in host memory)" % (dt, nseg * nsweeps / dt)
    if mode != "host":
        t0 = time.perf_counter()
        ctx.loam_run_packed(resident)
        dr = time.perf_counter() - t0
        msg += " in %.4f s = %.0f sweeps/s (sweeps in HBM)" % (dr, nseg * nsweeps / dr)
    print(msg, flush=True)
