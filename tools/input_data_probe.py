"""gpscal_input_data_run alone on the bench's bag -> KML input (2 bags x 100 sweeps of 13.9k points, long / short /
overlap 50 / 22 / 8 m), for timing and rocprofv3 kernel traces: python tools/input_data_probe.py [nbag] [nsweeps] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpscalibration_amd import Context, synth
nbag = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nsweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
W = synth.lidar_world(0, length=0.8 * nsweeps * nbag + 200.0)
bags, stamps = [], []
for b in range(nbag):
    sw, st, _ = synth.drive(W, nsweeps, seed=40 + b, n_az=900, start=(0.8 * nsweeps * b, 0.0))
    bags.append(sw); stamps.append(st + 0.1 * nsweeps * b)
ctx = Context(0)
ctx.input_data_run([bags[0][:6]], [stamps[0][:6]], 50.0, 22.0, 8.0)
for r in range(reps):
    t0 = time.perf_counter()
    tr = ctx.input_data_run(bags, stamps, 50.0, 22.0, 8.0)
    dt = time.perf_counter() - t0
    print("run %d: %d bags x %d sweeps -> %d tracks in %.4f s (%.2f ms per replay step of both passes)" % (
        r, nbag, nsweeps, len(tr), dt, 1e3 * dt / nsweeps), flush=True)
