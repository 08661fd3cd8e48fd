"""Two index builds of the benchmark batch (the second with warm caches) for a kernel trace:
rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/build_run.py [pairs points]; then
python tools/build_timeline.py <dir>"""
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
for k in range(3):
    torch.cuda.synchronize()
    time.sleep(0.02)
    t0 = time.perf_counter()
    sb = ctx.scan_batch(d_tg, to, d_sr, so)
    ctx.sync()
    print("build %d: %.3f ms (build_seconds %.3f ms)" % (k, 1e3 * (time.perf_counter() - t0), 1e3 * sb.build_seconds), flush=True)
    sb.close()
