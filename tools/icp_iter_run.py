"""Runs N ICP iterations on the benchmark batch (for per-dispatch rocprofv3 --pmc passes)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from gpscalibration_amd import Context, synth
npairs, n, iters = 64, 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 12
tg, to, sr, so, _ = synth.scan_batch(npairs, n)
ctx = Context(0)
sb = ctx.scan_batch(torch.from_numpy(tg).cuda(), to, torch.from_numpy(sr).cuda(), so)
sb.icp(iters, want_err=False)
ctx.sync()
