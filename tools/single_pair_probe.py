"""Single-pair ICP rate (the latency-bound case: one 65 536-point pair, 50 iterations per run, graph replay)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from gpscalibration_amd import Context, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = Context(0)
tg, to, sr, so, _ = synth.scan_batch(1, n)
sb = ctx.scan_batch(torch.from_numpy(tg).cuda(), to, torch.from_numpy(sr).cuda(), so)
d_T = torch.empty((1, 4, 4), dtype=torch.float64, device="cuda")
for rep in range(3):
    sb.set_pose(None); sb.icp(iters, want_err=False, T_out=d_T); ctx.sync()
    t1 = time.perf_counter()
    for _ in range(20):
        sb.set_pose(None)
        sb.icp(iters, want_err=False, T_out=d_T)
    ctx.sync()
    dt = time.perf_counter() - t1
    print("single pair: %.0f iterations/s (%.1f us per iteration)" % (20 * iters / dt, 1e6 * dt / (20 * iters)), flush=True)
sb.set_pose(None)
_, _, ms = sb.icp(iters, want_err=False, profile=True)
print("step launches (us):", " ".join("%.1f" % (1e3 * v) for v in ms[:16]), "... mean %.1f" % (1e3 * ms.mean()))
sb.close(); ctx.close()
