import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpscalibration_amd import Context, synth
npairs, n = int(sys.argv[1]), int(sys.argv[2])
ctx = Context(0)
tg, to, sr, so, Tt = synth.scan_batch(npairs, n)
d_tg = torch.from_numpy(tg).cuda(); d_sr = torch.from_numpy(sr).cuda()
torch.cuda.synchronize()
wt, wo, ws, wso, _ = synth.scan_batch(max(npairs, 1), 512)
ctx.scan_batch(wt, wo, ws, wso).close()
for i in range(4):
    t0 = time.perf_counter()
    sb = ctx.scan_batch(d_tg, to, d_sr, so)
    t1 = time.perf_counter()
    print("build %d: build_seconds %.6f wall %.6f" % (i, sb.build_seconds, t1 - t0), flush=True)
    sb.close()
ctx.close()
