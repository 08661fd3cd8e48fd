"""Runs odd-shaped inputs through the C ABI, each case in its own process: a case must end in a result or a
gpscal error code -- never a GPU fault, an abort or a hang.  tools/edge_probe.py [case ...]"""
import os, subprocess, sys
sys.path.insert(0, "/root/repo")
import numpy as np

CASES = {}


def case(f):
    CASES[f.__name__] = f
    return f


def _ctx():
    from gpscalibration_amd import Context
    return Context(0)


def _rng():
    return np.random.default_rng(3)


@case
def knn_empty_queries():
    c = _ctx(); ix = c.knn_index(_rng().normal(0, 1, (100, 3)).astype(np.float32))
    i, d = ix.search(np.zeros((0, 3), np.float32), 1); assert i.shape[0] == 0


@case
def knn_k_larger_than_cloud():
    c = _ctx(); ix = c.knn_index(_rng().normal(0, 1, (3, 3)).astype(np.float32))
    i, d = ix.search(_rng().normal(0, 1, (10, 3)).astype(np.float32), 8); assert (i[:, 3:] == -1).all()


@case
def knn_all_nan_target():
    c = _ctx(); ix = c.knn_index(np.full((50, 3), np.nan, np.float32))
    i, d = ix.search(_rng().normal(0, 1, (10, 3)).astype(np.float32), 2); assert (i == -1).all()


@case
def knn_huge_coordinates():
    c = _ctx(); t = (_rng().normal(0, 1, (500, 3)) * 1e6).astype(np.float32)
    ix = c.knn_index(t); i, d = ix.search(t[:50], 1); assert (i[:, 0] == np.arange(50)).all()


@case
def batch_empty_source_pair():
    c = _ctx(); r = _rng(); tg = r.normal(0, 1, (600, 3)).astype(np.float32)
    to = np.array([0, 300, 600], np.int64); sr = r.normal(0, 1, (200, 3)).astype(np.float32)
    so = np.array([0, 0, 200], np.int64)
    sb = c.scan_batch(tg, to, sr, so); T, e, _ = sb.icp(3); assert np.allclose(T[0], np.eye(4))


@case
def batch_empty_target_pair():
    c = _ctx(); r = _rng(); tg = r.normal(0, 1, (300, 3)).astype(np.float32)
    to = np.array([0, 0, 300], np.int64); sr = r.normal(0, 1, (400, 3)).astype(np.float32)
    so = np.array([0, 200, 400], np.int64)
    sb = c.scan_batch(tg, to, sr, so); T, e, _ = sb.icp(3); assert np.allclose(T[0], np.eye(4))
    i, d = sb.correspondences(); assert (i[:200] == -1).all()


@case
def batch_everything_empty():
    c = _ctx()
    sb = c.scan_batch(np.zeros((1, 3), np.float32)[:0], np.array([0, 0], np.int64), np.zeros((1, 3), np.float32)[:0], np.array([0, 0], np.int64))
    T, e, _ = sb.icp(2); assert np.allclose(T[0], np.eye(4))


@case
def batch_nan_source_is_an_error():
    from gpscalibration_amd._lib import GpscalError
    c = _ctx(); r = _rng(); tg = r.normal(0, 1, (300, 3)).astype(np.float32); sr = r.normal(0, 1, (100, 3)).astype(np.float32)
    sr[7, 1] = np.nan
    try:
        c.scan_batch(tg, np.array([0, 300], np.int64), sr, np.array([0, 100], np.int64)); raise SystemExit("no error")
    except GpscalError:
        pass


@case
def batch_weighted_small():
    c = _ctx(); r = _rng(); tg = r.normal(0, 1, (300, 3)).astype(np.float32); sr = r.normal(0, 1, (37, 3)).astype(np.float32)
    sb = c.scan_batch(tg, np.array([0, 300], np.int64), sr, np.array([0, 37], np.int64), w=r.uniform(0.1, 1, 37))
    sb.icp(4)


@case
def batch_zero_iterations():
    c = _ctx(); r = _rng(); tg = r.normal(0, 1, (300, 3)).astype(np.float32)
    sb = c.scan_batch(tg, np.array([0, 300], np.int64), tg[:50], np.array([0, 50], np.int64)); T, e, _ = sb.icp(0)


@case
def sr_tiny_and_empty_sweeps():
    from gpscalibration_amd._lib import GpscalError
    c = _ctx(); r = _rng()
    for n in (0, 1, 10, 100):
        try:
            out = c.scan_registration([r.normal(0, 10, (n, 3)).astype(np.float32)])
        except GpscalError as e:
            print("  n=%d -> error %s" % (n, e))


@case
def sr_all_points_one_ring():
    c = _ctx(); a = np.linspace(0, 2 * np.pi, 3000, endpoint=False)
    sw = np.stack([10 * np.cos(a), 10 * np.sin(a), np.zeros_like(a)], 1).astype(np.float32)
    c.scan_registration([sw])


@case
def sr_nan_and_zero_points():
    c = _ctx(); r = _rng(); sw = r.normal(0, 10, (5000, 3)).astype(np.float32); sw[::7] = np.nan; sw[::11] = 0
    c.scan_registration([sw])


@case
def voxel_empty_single_huge_leaf():
    c = _ctx(); r = _rng()
    out = c.voxel_grid([np.zeros((0, 4), np.float32), r.normal(0, 1, (1, 4)).astype(np.float32), r.normal(0, 5, (5000, 4)).astype(np.float32)], 1e6)
    print("  kept", [len(o) for o in out])
    assert len(out[0]) == 0 and len(out[1]) == 1 and 1 <= len(out[2]) <= 8


@case
def voxel_tiny_leaf_is_an_error_or_identity():
    from gpscalibration_amd._lib import GpscalError
    c = _ctx(); r = _rng()
    try:
        out = c.voxel_grid([r.normal(0, 50, (2000, 4)).astype(np.float32)], 1e-6); print("  kept", len(out[0]))
    except GpscalError as e:
        print("  error", e)


@case
def loam_run_short_segments():
    from gpscalibration_amd import synth
    from gpscalibration_amd._lib import GpscalError
    c = _ctx(); W = synth.lidar_world(0, length=200.0); sw, st, _ = synth.drive(W, 6, seed=1, n_az=600)
    for k in (1, 2, 3):
        try:
            out = c.loam_run([sw[:k], sw[:k + 1]], [st[:k], st[:k + 1]]); assert len(out[0]["track"]) == k
        except GpscalError as e:
            print("  k=%d error %s" % (k, e))


@case
def loam_run_featureless_sweeps():
    from gpscalibration_amd._lib import GpscalError
    c = _ctx(); r = _rng()
    a = np.linspace(0, 2 * np.pi, 8000, endpoint=False)
    ring = np.stack([20 * np.cos(a), 20 * np.sin(a), 0 * a], 1).astype(np.float32)  # one ring only: no features elsewhere
    try:
        c.loam_run([[ring, ring, ring, ring]], [np.arange(4) * 0.1])
    except GpscalError as e:
        print("  error", e)


@case
def loam_run_random_noise_sweeps():
    from gpscalibration_amd._lib import GpscalError
    c = _ctx(); r = _rng()
    sws = [r.normal(0, 15, (20000, 3)).astype(np.float32) for _ in range(5)]
    try:
        c.loam_run([sws], [np.arange(5) * 0.1])
    except GpscalError as e:
        print("  error", e)


@case
def input_data_tiny_bags():
    from gpscalibration_amd import synth
    from gpscalibration_amd._lib import GpscalError
    c = _ctx(); W = synth.lidar_world(0, length=200.0); sw, st, _ = synth.drive(W, 5, seed=1, n_az=600)
    for k in (1, 2, 5):
        try:
            out = c.input_data_run([sw[:k]], [st[:k]], 50.0, 22.0, 8.0); print("  k=%d tracks %d" % (k, len(out)))
        except GpscalError as e:
            print("  k=%d error %s" % (k, e))


@case
def track_fit_degenerate_segments():
    from gpscalibration_amd._lib import GpscalError
    c = _ctx(); r = _rng()
    for n in (1, 2):
        s = np.zeros((n, 4)); s[:, 3] = np.arange(n); e = s.copy(); w = np.ones(n)
        try:
            c.track_fit(s, e, w)
        except GpscalError as ex:
            print("  n=%d error %s" % (n, ex))
    s = np.zeros((50, 4)); s[:, 3] = np.arange(50)  # a standing vehicle: all poses identical
    c.track_fit(s, s.copy(), np.ones(50)); c.long_segment(s, s.copy())


@case
def odometry_empty_feature_sets():
    c = _ctx(); r = _rng(); z = np.zeros((0, 4), np.float32); p = r.normal(0, 10, (500, 4)).astype(np.float32)
    c.loam_odometry([z], [z], [p], [p]); c.loam_odometry([p[:50]], [p[:80]], [z], [z])
    c.loam_mapping([z], [z], [p], [p]); c.loam_mapping([p[:50]], [p[:80]], [z], [z])


@case
def sr_maximum_size_single_ring_and_random():
    from gpscalibration_amd._lib import GpscalError
    c = _ctx(); r = _rng()
    a = np.linspace(0, 2 * np.pi, 59000, endpoint=False)
    one_ring = np.stack([10 * np.cos(a), 10 * np.sin(a), np.zeros_like(a)], 1).astype(np.float32)
    two_rings = one_ring.copy(); two_rings[::2, 2] = 10 * np.tan(np.deg2rad(7.0))
    rnd = r.normal(0, 20, (60000, 3)).astype(np.float32)
    for name, sw in (("one ring", one_ring), ("two rings", two_rings), ("random 60000", rnd), ("random 60001", np.concatenate([rnd, rnd[:1]]))):
        try:
            out = c.scan_registration([sw]); print("  %s: %s" % (name, {k: len(v) for k, v in out[0].items()}))
        except GpscalError as e:
            print("  %s -> error %s" % (name, e))


@case
def knn_k_out_of_range_is_an_error():
    from gpscalibration_amd._lib import GpscalError
    c = _ctx(); ix = c.knn_index(_rng().normal(0, 1, (100, 3)).astype(np.float32))
    for k in (0, 9, -1):
        try:
            ix.search(np.zeros((4, 3), np.float32), k); raise SystemExit("k=%d accepted" % k)
        except (GpscalError, ValueError) as e:
            print("  k=%d -> %s" % (k, str(e)[:80]))


@case
def track_long_and_many_segments():
    c = _ctx(); r = _rng()
    n = 20000
    s = np.cumsum(r.normal(0, 1, (n, 4)), 0); s[:, 3] = np.arange(n) * 0.1
    e = s + r.normal(0, 0.5, (n, 4)); e[:, 3] = s[:, 3]
    c.track_fit(s, e, np.ones(n)); c.long_segment(s, e)
    off = np.arange(0, n + 1, 2).astype(np.int32)  # 10 000 segments of two poses
    c.track_fit(s, e, np.ones(n), seg_offsets=off); c.long_segment(s, e, seg_offsets=off)


@case
def voxel_large_cloud_many_cells():
    c = _ctx(); r = _rng()
    out = c.voxel_grid([r.uniform(-200, 200, (300000, 4)).astype(np.float32)], 0.5); print("  kept", len(out[0]))


if len(sys.argv) > 2 and sys.argv[1] == "--run":
    CASES[sys.argv[2]]()
    print("  done")
    sys.exit(0)
names = sys.argv[1:] or list(CASES)
bad = 0
for n in names:
    try:
        r = subprocess.run([sys.executable, __file__, "--run", n], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
        out, rc = r.stdout.decode(), r.returncode
    except subprocess.TimeoutExpired as e:
        out, rc = (e.stdout or b"").decode(), "TIMEOUT"
    verdict = "ok" if rc == 0 else ("FAULT" if "Memory access fault" in out else "FAILED rc=%s" % rc)
    bad += verdict != "ok"
    print("%-40s %s" % (n, verdict), flush=True)
    keep = [l for l in out.splitlines() if l.startswith("  ") and not l.startswith("  File")] if rc == 0 else out.splitlines()[-12:]
    for l in keep:
        print("     " + l[:200], flush=True)
sys.exit(1 if bad else 0)
