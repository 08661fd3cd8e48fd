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


_KML_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["GPSCAL_ROOT"])
from gpscalibration_amd import Context, pipeline, synth
from gpscalibration_amd.parallel import bag_to_kml_sharded, gather_doubles_dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
out = os.environ["GPSCAL_OUT"]
W = synth.lidar_world(0, length=400.0)
bags, stamps, xy = [], [], []
for b, n in enumerate((60, 44, 52)):
    sw, st, truth = synth.drive(W, n, seed=70 + b, n_az=450, start=(48.0 * b, 0.0))
    bags.append(sw); stamps.append(st + 6.0 * b); xy.append(truth[:, :2])
log = out + ".gps.txt"
if rank == 0:
    open(log, "w").write(synth.gprmc_for_path(np.concatenate(stamps), np.concatenate(xy), seed=5, sigma=1.0))
dist.barrier()
ctx = Context(0)
L, S, OV = 20.0, 9.0, 3.0
slam = lambda b, s: ctx.input_data_run(b, s, L, S, OV)
tracks = lambda g, lo, sh, k0, k1: pipeline.run_tracks(g, lo, sh, kml_original=k0, kml_calibrated=k1)
r = bag_to_kml_sharded(bags, stamps, log, slam, tracks, rank, world, gather_doubles_dist(dist), out + ".ori.kml", out + ".cal.kml")
if rank == 0:
    r1 = bag_to_kml_sharded(bags, stamps, log, slam, tracks, 0, 1, None, out + ".ref.ori.kml", out + ".ref.cal.kml")
    assert r1["segments"] == r["segments"] and r["segments"][0] >= 3 and r["segments"][1] >= 6, r["segments"]
    for ext in (".ori.kml", ".cal.kml"):
        a, b = open(out + ext, "rb").read(), open(out + ".ref" + ext, "rb").read()
        assert len(a) > 500 and a == b, ext
dist.barrier()
dist.destroy_process_group()
ctx.close()
print("rank %d ok" % rank)
'''


def test_bag_to_kml_sharded_over_two_ranks(tmp_path):
    """BASELINE configs[3] / [4] shape on the one-GPU box: three bags over two ranks (both on GPU 0, exchange over
    gloo), each rank runs input_data's replay + the LOAM nodes on its bags, ONE ragged exchange of the segments'
    pose chains, global long / short passes + merge + KML on rank 0.  The files must equal the single-process
    run byte for byte."""
    script = tmp_path / "worker.py"
    script.write_text(_KML_WORKER)
    env = dict(os.environ, GPSCAL_ROOT=ROOT, GPSCAL_OUT=str(tmp_path / "kml"), MASTER_ADDR="127.0.0.1", MASTER_PORT="29537")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29537", str(script)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def test_rccl_allgather_chains_ragged_path_world_of_one(monkeypatch):
    """The exported RCCL exchange (gpscal_comm_init / gpscal_allgather_chains) on the one-GPU box: the ragged
    branch (grouped ncclBroadcast, in place) forced by GPSCAL_COMM_FORCE_RAGGED, three different chain sizes,
    host and device buffers.  N > 1 runs only in the driver's multi-GPU bench (bench.py routes its gather
    through this entry point); until then it is unverified beyond one rank."""
    from gpscalibration_amd import Context, GpscalError
    ctx = Context(0)
    try:
        ctx.comm_init(Context.comm_unique_id(), 0, 1)
    except GpscalError as e:
        ctx.close()
        pytest.skip("RCCL not usable here: %s" % e)
    try:
        _rccl_world_of_one_body(ctx, monkeypatch)
    finally:
        ctx.comm_destroy()
        ctx.close()


def _rccl_world_of_one_body(ctx, monkeypatch):
    from gpscalibration_amd import GpscalError
    rng = np.random.default_rng(3)
    for force in (False, True):
        if force:
            monkeypatch.setenv("GPSCAL_COMM_FORCE_RAGGED", "1")
        for n in (4, 4 * 1237, 4 * 100003):
            local = rng.normal(size=n)
            got = ctx.allgather_chains(local, [n])
            assert np.array_equal(got, local), (force, n)
        assert len(ctx.allgather_chains(np.zeros(0), [0])) == 0
    import torch
    d_in = torch.from_numpy(rng.normal(size=4096)).cuda()
    d_out = torch.empty(4096, dtype=torch.float64, device="cuda")
    from gpscalibration_amd.api import _ptr
    cnt = np.array([4096], dtype=np.int32)
    ctx._ck(ctx._L.gpscal_allgather_chains(ctx._h, _ptr(d_in), _ptr(cnt), _ptr(d_out)), "allgather device")
    assert torch.equal(d_in, d_out)
    with pytest.raises(GpscalError):  # NULL local with a non-zero count
        ctx._ck(ctx._L.gpscal_allgather_chains(ctx._h, None, _ptr(cnt), _ptr(d_out)), "allgather null")
    # device pointers in and out: the gather is enqueued behind whatever the context's stream holds and the call
    # returns without waiting for it (host pointers make it wait, as everywhere in the ABI)
    import time
    from gpscalibration_amd import synth
    npairs = 24
    tg, to, sr, so, _ = synth.scan_batch(npairs, 65536)
    sb = ctx.scan_batch(torch.from_numpy(tg).cuda(), to, torch.from_numpy(sr).cuda(), so)
    d_T = torch.empty((npairs, 4, 4), dtype=torch.float64, device="cuda")
    d_all = torch.empty((npairs, 4, 4), dtype=torch.float64, device="cuda")
    cnts = np.array([16 * npairs], dtype=np.int32)
    sb.icp(50, want_err=False, T_out=d_T)  # warm-up: graph capture
    ctx.sync()
    for _ in range(16):  # 16 graph replays of 24 x 50 iterations queued: milliseconds of work still pending ...
        sb.icp(50, want_err=False, T_out=d_T)  # (no set_pose here: a host pose makes that call wait for the stream)
    t1 = time.perf_counter()
    ctx._ck(ctx._L.gpscal_allgather_chains(ctx._h, _ptr(d_T), _ptr(cnts), _ptr(d_all)), "allgather behind work")
    t_call = time.perf_counter() - t1
    ctx.sync()
    t_drain = time.perf_counter() - t1
    assert t_call < 0.25 * t_drain, (t_call, t_drain)  # ... which the gather did not wait for
    torch.cuda.synchronize()
    assert torch.equal(d_T, d_all)
    sb.close()


_COEXIST = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["GPSCAL_ROOT"])
from gpscalibration_amd import Context
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda")
dist.all_reduce(t)                       # torch's RCCL communicator is live
ctx = Context(0)
uid = [Context.comm_unique_id()]
dist.broadcast_object_list(uid, src=0)   # the way bench.py hands the id around
ctx.comm_init(uid[0], 0, 1)
x = np.arange(4096, dtype=np.float64)
assert np.array_equal(ctx.allgather_chains(x, [4096]), x)
d_in = torch.arange(64 * 16, dtype=torch.float64, device="cuda")
d_out = torch.empty_like(d_in)
from gpscalibration_amd.api import _ptr
cnt = np.array([64 * 16], dtype=np.int32)
for _ in range(3):
    ctx._ck(ctx._L.gpscal_allgather_chains(ctx._h, _ptr(d_in), _ptr(cnt), _ptr(d_out)), "gather")
    dist.all_reduce(t)                   # interleaved with torch's collectives
assert torch.equal(d_in, d_out)
ctx.comm_destroy(); ctx.close()
dist.barrier(); dist.destroy_process_group()
print("coexist ok")
'''


def test_library_rccl_communicator_next_to_torch_distributed(tmp_path):
    """bench.py --gpus N creates the library's RCCL communicator (gpscal_comm_init, librccl dlopen'ed by the library)
    inside a process whose torch.distributed NCCL backend is already up, and interleaves both.  One rank, one GPU:
    what can be checked here is that the two coexist and that the id travels through broadcast_object_list."""
    script = tmp_path / "coexist.py"
    script.write_text(_COEXIST)
    env = dict(os.environ, GPSCAL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "coexist ok" in r.stdout, r.stdout[-3000:]
