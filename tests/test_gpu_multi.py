"""N > 1 path of the LOAM chain on the one-GPU box: two processes (ranks) share the card, each runs
its contiguous block of segments, the tracks are exchanged with torch.distributed (gloo here: RCCL
refuses two ranks on one device; the driver's 8-GPU run uses backend nccl = RCCL).  The gathered
tracks must equal a single-process run bit for bit -- segments never interact."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["GPSCAL_ROOT"])
from gpscalibration_amd import Context, synth
from gpscalibration_amd.parallel import loam_run_sharded
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
W = synth.lidar_world(0)
segs, stamps = [], []
for s in range(3):
    sw, st, _ = synth.drive(W, 8 + 2 * s, seed=20 + s, n_az=450, start=(30.0 * s, 0.2 * s))
    segs.append(sw); stamps.append(st)
ctx = Context(0)
tracks = loam_run_sharded(ctx, segs, stamps, dist)
ref = loam_run_sharded(ctx, segs, stamps, None) if rank == 0 else None
if rank == 0:
    assert len(tracks) == 3
    for a, b in zip(tracks, ref):
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
    assert all(np.isnan(t[0]).all() and np.isfinite(t[1:]).all() for t in tracks)
dist.barrier()
dist.destroy_process_group()
ctx.close()
print("rank %d ok" % rank)
'''


def test_loam_chain_sharded_over_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, GPSCAL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout
