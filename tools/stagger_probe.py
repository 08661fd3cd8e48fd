"""Do the search-heavy iterations (bound by per-lane gathers) and the converged iterations (bound by streaming and
latency) of icp_step_kernel overlap well when they run at the same time?  Two contexts (two streams), half the pairs
each: (a) one batch of all pairs, (b) two half batches started together, (c) the half batches half a run out of phase
(B's search iterations while A runs its converged ones and vice versa), enforced with cross-stream waits."""
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
head = int(sys.argv[4]) if len(sys.argv) > 4 else 8
steps = 10
tg, to, sr, so, _ = synth.scan_batch(npairs, n)
d_tg, d_sr = torch.from_numpy(tg).cuda(), torch.from_numpy(sr).cuda()
h = npairs // 2


def run(fn, sync):
    fn(); fn(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


ctx = Context(0)
sb = ctx.scan_batch(d_tg, to, d_sr, so)


def one():
    sb.set_pose(None); sb.icp(iters, want_err=False)


dt = run(one, ctx.sync)
print("one batch of %d pairs: %.3f ms per step, %.0f k it/s" % (npairs, 1e3 * dt, npairs * iters / dt / 1e3), flush=True)
T_one = sb.icp(0)[0] if False else None
sb.close()

A, Bc = Context(0), Context(0)
offA = to[:h + 1].copy(); offB = (to[h:] - to[h]).copy()
soA = so[:h + 1].copy(); soB = (so[h:] - so[h]).copy()
sbA = A.scan_batch(d_tg[:to[h]], offA, d_sr[:so[h]], soA)
sbB = Bc.scan_batch(d_tg[to[h]:], offB, d_sr[so[h]:], soB)


def both_sync():
    A.sync(); Bc.sync()


def in_phase():
    sbA.set_pose(None); sbA.icp(iters, want_err=False)
    sbB.set_pose(None); sbB.icp(iters, want_err=False)


dt = run(in_phase, both_sync)
print("two half batches, started together: %.3f ms per step, %.0f k it/s" % (1e3 * dt, npairs * iters / dt / 1e3), flush=True)

L = A._L
state = {"primed": False}


def staggered():
    if not state["primed"]:
        sbA.set_pose(None); sbA.icp(head, want_err=False)
        state["primed"] = True
    # B's search iterations wait for A's, then run next to A's converged ones
    L.gpscal_wait_for_stream(Bc._h, A.stream)
    sbA.icp(iters - head, want_err=False)
    sbB.set_pose(None); sbB.icp(head, want_err=False)
    # A's next search iterations wait for B's, then run next to B's converged ones
    L.gpscal_wait_for_stream(A._h, Bc.stream)
    sbB.icp(iters - head, want_err=False)
    sbA.set_pose(None); sbA.icp(head, want_err=False)


dt = run(staggered, both_sync)
print("two half batches, half a run out of phase (head %d): %.3f ms per step, %.0f k it/s" % (head, 1e3 * dt, npairs * iters / dt / 1e3), flush=True)
