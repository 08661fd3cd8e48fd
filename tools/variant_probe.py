"""One library build (GPSCAL_LIB=variants/libgpscal_X.so) on the benchmark batch: per-iteration launch times of
icp_step_kernel, graph-replay throughput, and a checksum of poses + correspondences (variants must agree bit for bit)."""
import hashlib
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gpscalibration_amd import Context, synth
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
tg, to, sr, so, _ = synth.scan_batch(npairs, n)
ctx = Context(0)
sb = ctx.scan_batch(torch.from_numpy(tg).cuda(), to, torch.from_numpy(sr).cuda(), so)
sb.icp(2)
best = None
for _ in range(3):
    sb.set_pose(None)
    _, _, ms = sb.icp(iters, want_err=False, profile=True)
    best = ms if best is None else np.minimum(best, ms)
sb.set_pose(None)
T, err, _ = sb.icp(iters)
idx, sqd = sb.correspondences()
h = hashlib.sha256(np.ascontiguousarray(T).tobytes() + np.ascontiguousarray(idx).tobytes() +
                   np.ascontiguousarray(sqd).tobytes()).hexdigest()[:16]
for _ in range(2):
    sb.set_pose(None); sb.icp(iters, want_err=False)
ctx.sync()
t0 = time.perf_counter()
steps = 5
for _ in range(steps):
    sb.set_pose(None); sb.icp(iters, want_err=False)
ctx.sync()
dt = (time.perf_counter() - t0) / steps
us = 1e3 * best
print("%s | %s | sum %.0f us mean %.1f us (frac %.3f) search(2-8) %.0f converged(last 30) %.1f | %.0f k it/s | sha %s" % (
    os.environ.get("GPSCAL_LIB", "default"), " ".join("%.0f" % v for v in us[:16]), us.sum(), us.mean(),
    npairs * 32.0 * n / (us.mean() * 1e-6) / 8e12, us[1:8].sum(), us[-30:].mean(), npairs * iters / dt / 1e3, h), flush=True)
