"""The bench's timed loop (set_pose + icp with device outputs) with the schedule's decisions printed
(GPSCAL_SCHED_DEBUG=1 python tools/sched_probe.py)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from gpscalibration_amd import Context, synth
npairs, n, iters = 64, 65536, 50
ctx = Context(0)
tg, to, sr, so, _ = synth.scan_batch(npairs, n)
sb = ctx.scan_batch(torch.from_numpy(tg).cuda(), to, torch.from_numpy(sr).cuda(), so)
d_T = torch.empty((npairs, 4, 4), dtype=torch.float64, device="cuda")
d_err = torch.empty((npairs, iters), dtype=torch.float64, device="cuda")
for mode in ("err", "noerr", "err", "noerr"):
    for rep in range(2):
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(10):
            sb.set_pose(None)
            if mode == "err":
                sb.icp(iters, T_out=d_T, err_out=d_err)
            else:
                sb.icp(iters, want_err=False, T_out=d_T)
        ctx.sync(); dt = (time.perf_counter() - t0) / 10
        print("%s: %.3f ms per step -> %.0f k it/s" % (mode, 1e3 * dt, npairs * iters / dt / 1e3), flush=True)
