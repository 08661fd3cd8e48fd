"""Does the replayed 4-chain graph of a batch have 'modes'?  Several batches over the same data in one process, each
with its own captured graph: time of 10 replays of 50 iterations each, twice per batch.
python tools/graph_mode_probe.py [pairs points]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gpscalibration_amd import Context, synth
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
tg, to, sr, so, _ = synth.scan_batch(npairs, n)
ctx = Context(0)
d_tg, d_sr = torch.from_numpy(tg).cuda(), torch.from_numpy(sr).cuda()
d_T = torch.empty((npairs, 4, 4), dtype=torch.float64, device="cuda")
d_err = torch.empty((npairs, 50), dtype=torch.float64, device="cuda")
for trial in range(6):
    sb = ctx.scan_batch(d_tg, to, d_sr, so)
    out = []
    for rep in range(3):
        sb.set_pose(None)
        sb.icp(50, T_out=d_T, err_out=d_err)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            sb.set_pose(None)
            sb.icp(50, T_out=d_T, err_out=d_err)
        ctx.sync()
        out.append((time.perf_counter() - t0) / 10 * 1e3)
    print("batch %d: %s ms per 50 iterations" % (trial, " ".join("%.3f" % x for x in out)), flush=True)
    sb.close()
