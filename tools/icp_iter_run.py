"""Runs N ICP iterations on the benchmark batch as whole-batch launches of icp_step_kernel (profiling mode of
gpscal_scan_batch_icp: the launch the bench's roofline is quoted for), for per-dispatch rocprofv3 --pmc passes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gpscalibration_amd import Context, synth
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
tg, to, sr, so, _ = synth.scan_batch(npairs, n)
ctx = Context(0)
sb = ctx.scan_batch(torch.from_numpy(tg).cuda(), to, torch.from_numpy(sr).cuda(), so)
sb.icp(iters, want_err=True)  # a first run (graph capture, caches warm)
sb.set_pose(None)
sb.icp(iters, want_err=True, profile=True)
ctx.sync()
