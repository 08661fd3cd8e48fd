"""GPU parity tests: the HIP path, called through the C ABI (libgpscal_hip.so), against
the CPU oracle on the same seeded inputs.  Bars:
  * k-NN indices and squared distances: BIT-EXACT (integer / index work; both sides
    compute fmaf(dz,dz,fmaf(dy,dy,dx*dx)) and order by (d2, index));
  * one ICP iteration from the same pose: pose within 1e-9 (float64 reductions in a
    different summation order);
  * multi-iteration ICP: pose within 1e-5 (float32 rounding of the running pose may flip
    individual correspondences);
  * track path (float64): positions within 1e-6 m, weights within 1e-6 relative -- the
    easting is ~4e8 m, where one float64 ulp is 6e-8 m; KML degrees within 1e-9
    (north_star: 1e-6 deg);
  * projections: within 1e-6 m / 1e-10 deg of the oracle (device libm vs glibc).
"""
import math

import numpy as np
import pytest

import _oracle as O
from gpscalibration_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from gpscalibration_amd import Context
    c = Context(0)
    yield c
    c.close()


# -------------------------------------------------------------------- k-NN
@pytest.mark.parametrize("k", [1, 5])
@pytest.mark.parametrize("m,n", [(4096, 3000), (20000, 7777)])
def test_knn_bit_exact_vs_oracle(ctx, k, m, n):
    tgt = synth.scan_scene(m, 11)
    rng = np.random.default_rng(5)
    q = synth.scan_scene(n, 12) + rng.normal(0, 0.3, size=(n, 3)).astype(np.float32)
    q[:100] = tgt[:100]  # exact hits
    q[100:110] = np.array([500.0, -300.0, 80.0], dtype=np.float32)  # far outside the grid
    ix = ctx.knn_index(tgt)
    idx, sqd = ix.search(q, k)
    kd = O.KdTree(tgt)
    ridx, rsqd = kd.search(q, k)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(sqd, rsqd)
    ix.close()


def test_knn_ties_duplicates_and_small_clouds(ctx):
    rng = np.random.default_rng(1)
    tgt = (rng.normal(size=(1000, 3)) * 5).astype(np.float32)
    tgt[500:600] = tgt[0:100]  # duplicates: the lower index must win
    q = tgt[:200].copy()
    ix = ctx.knn_index(tgt)
    idx, sqd = ix.search(q, 2)
    ridx, rsqd = O.knn_brute(tgt, q, 2)
    assert np.array_equal(idx, ridx) and np.array_equal(sqd, rsqd)
    assert np.all(idx[:100, 0] == np.arange(100)) and np.all(idx[:100, 1] == np.arange(500, 600))
    ix.close()
    # k larger than the cloud: missing neighbours are -1 / inf
    small = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], dtype=np.float32)
    ix = ctx.knn_index(small)
    idx, sqd = ix.search(np.array([[0.4, 0, 0]], dtype=np.float32), 5)
    assert list(idx[0]) == [0, 1, 2, -1, -1] and np.isinf(sqd[0, 3])
    ix.close()
    # single point, coplanar and collinear clouds
    ix = ctx.knn_index(small[:1])
    idx, sqd = ix.search(np.array([[3, 4, 0]], dtype=np.float32), 1)
    assert idx[0, 0] == 0 and sqd[0, 0] == 25.0
    ix.close()
    line = np.c_[np.linspace(0, 100, 5000), np.zeros(5000), np.zeros(5000)].astype(np.float32)
    ix = ctx.knn_index(line)
    qq = (rng.uniform(-10, 110, size=(2000, 3)) * np.array([1, 0.05, 0.05])).astype(np.float32)
    idx, sqd = ix.search(qq, 3)
    ridx, rsqd = O.knn_brute(line, qq, 3)
    assert np.array_equal(idx, ridx) and np.array_equal(sqd, rsqd)
    ix.close()


def test_knn_strided_pointxyzi_and_nan_points(ctx):
    # pcl::PointXYZI is 32 bytes; NaN returns are skipped like
    # removeNaNFromPointCloud does upstream (scanRegistration.cpp:260-263)
    rng = np.random.default_rng(2)
    m = 3000
    raw = np.zeros((m, 8), dtype=np.float32)
    raw[:, :3] = rng.normal(size=(m, 3)) * 8
    raw[:, 4] = rng.uniform(0, 255, m)
    raw[7, :3] = np.nan
    ix = ctx.knn_index(raw, stride_bytes=32)
    q = np.zeros((500, 8), dtype=np.float32)
    q[:, :3] = rng.normal(size=(500, 3)) * 8
    q[3, :3] = np.nan
    idx, sqd = ix.search(q, 1, stride_bytes=32)
    good = np.ones(m, dtype=bool)
    good[7] = False
    ridx, rsqd = O.knn_brute(raw[good, :3], q[:, :3], 1)
    remap = np.flatnonzero(good)
    ok = np.ones(500, dtype=bool)
    ok[3] = False
    assert np.array_equal(idx[ok, 0], remap[ridx[ok, 0]])
    assert np.array_equal(sqd[ok, 0], rsqd[ok, 0])
    assert idx[3, 0] == -1 and np.isinf(sqd[3, 0])
    ix.close()


def test_knn_device_pointers(ctx):
    torch = pytest.importorskip("torch")
    tgt = synth.scan_scene(8192, 3)
    q = synth.scan_scene(4096, 4)
    d_t = torch.from_numpy(tgt).cuda()
    d_q = torch.from_numpy(q).cuda()
    ix = ctx.knn_index(d_t)
    d_i = torch.empty((4096, 1), dtype=torch.int32, device="cuda")
    d_d = torch.empty((4096, 1), dtype=torch.float32, device="cuda")
    ix.search(d_q, 1, out_idx=d_i, out_sqd=d_d)
    ctx.sync()
    ridx, rsqd = O.KdTree(tgt).search(q, 1)
    assert np.array_equal(d_i.cpu().numpy(), ridx) and np.array_equal(d_d.cpu().numpy(), rsqd)
    ix.close()


def test_device_tensors_are_ordered_against_pending_torch_work(ctx):
    """include/gpscal.h: device-pointer arguments carry no ordering of their own.  The Python binding orders
    them (gpscal_wait_for_stream before the call, gpscal_make_stream_wait after it): a query tensor that a
    torch kernel queued behind milliseconds of other work is still writing, and result tensors read back
    through torch without any host synchronisation, must give the oracle's answer."""
    torch = pytest.importorskip("torch")
    tgt = synth.scan_scene(8192, 3)
    q = synth.scan_scene(4096, 4)
    d_t = torch.from_numpy(tgt).cuda()
    q_dev = torch.from_numpy(q).cuda()
    d_q = torch.zeros((4096, 3), dtype=torch.float32, device="cuda")  # wrong until the pending copy has run
    big = torch.randn(4096, 4096, device="cuda")
    ix = ctx.knn_index(d_t)
    d_i = torch.empty((4096, 1), dtype=torch.int32, device="cuda")
    d_d = torch.empty((4096, 1), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(30):  # milliseconds of queued work on torch's stream ...
        big = torch.tanh(big @ big)
    d_q.copy_(q_dev)  # ... and behind it the kernel that produces the query
    ix.search(d_q, 1, out_idx=d_i, out_sqd=d_d)  # no ctx.sync(), no torch.cuda.synchronize()
    got_i, got_d = d_i.cpu().numpy(), d_d.cpu().numpy()
    ridx, rsqd = O.KdTree(tgt).search(q, 1)
    assert np.array_equal(got_i, ridx) and np.array_equal(got_d, rsqd)
    ix.close()


# --------------------------------------------------------------------- ICP
def test_icp_single_iteration_matches_oracle(ctx):
    tgt, src, _ = synth.scan_pair(16384, 0)
    off = np.array([0, len(tgt)], dtype=np.int64)
    sb = ctx.scan_batch(tgt, off, src, off)
    T, err, _ = sb.icp(1)
    kd = O.KdTree(tgt)
    T_ref, e_ref, idx_ref, sqd_ref = kd.icp_iterate(src, np.eye(4))
    idx, sqd = sb.correspondences()
    assert np.array_equal(idx, idx_ref) and np.array_equal(sqd, sqd_ref)  # bit-exact correspondences
    assert np.abs(T[0] - T_ref).max() < 1e-9
    assert abs(err[0, 0] - e_ref) < 1e-10
    sb.close()


def test_icp_weighted_iteration_matches_oracle(ctx):
    tgt, src, _ = synth.scan_pair(8192, 3)
    w = np.random.default_rng(0).uniform(0.1, 2.0, len(src))
    off = np.array([0, len(tgt)], dtype=np.int64)
    sb = ctx.scan_batch(tgt, off, src, off, w=w)
    T0 = np.eye(4)
    T0[:3, :3] = synth.rot_zyx(0.5, 0.1, -0.2)
    T0[:3, 3] = [0.1, -0.2, 0.05]
    sb.set_pose(T0[None])
    T, err, _ = sb.icp(1)
    T_ref, e_ref, _, _ = O.KdTree(tgt).icp_iterate(src, T0, w)
    assert np.abs(T[0] - T_ref).max() < 1e-9
    assert abs(err[0, 0] - e_ref) < 1e-10
    sb.close()


def test_icp_batched_run_matches_oracle_and_truth(ctx):
    npairs, n = 6, 8192
    tg, to, sr, so, T_true = synth.scan_batch(npairs, n)
    sb = ctx.scan_batch(tg, to, sr, so)
    T, err, _ = sb.icp(30)
    for p in range(npairs):
        kd = O.KdTree(tg[to[p]:to[p + 1]])
        T_ref, hist = kd.icp_run(sr[so[p]:so[p + 1]], 30)
        assert np.abs(T[p] - T_ref).max() < 1e-5, p
        assert np.abs(err[p] - hist).max() < 1e-5
        assert np.abs(T[p][:3, :3] - T_true[p][:3, :3]).max() < 2e-3
        assert np.abs(T[p][:3, 3] - T_true[p][:3, 3]).max() < 0.05
    # replaying the captured graph from the same start pose is deterministic, bit for bit
    sb.set_pose(None)
    T2, err2, _ = sb.icp(30)
    assert np.array_equal(T, T2) and np.array_equal(err, err2)
    # profiling mode (eager, event-bracketed) computes the same thing
    sb.set_pose(None)
    T3, _, ms = sb.icp(30, profile=True)
    assert np.array_equal(T, T3) and ms.shape == (30,) and np.all(ms > 0)
    sb.close()


def test_icp_single_pair_api_over_prebuilt_index(ctx):
    tgt, src, _ = synth.scan_pair(4096, 1)
    ix = ctx.knn_index(tgt)
    T, hist = ix.icp_run(src, 10)
    T_ref, h_ref = O.KdTree(tgt).icp_run(src, 10)
    assert np.abs(T - T_ref).max() < 1e-5 and np.abs(hist - h_ref).max() < 1e-5
    ix.close()


def test_icp_full_size_properties(ctx):
    """BASELINE configs[1] size (65 536 points, 50 iterations): size-independent properties --
    convergence to the generating transform, monotone-ish error, idempotence at the fixed point."""
    tgt, src, T_true = synth.scan_pair(65536, 0)
    off = np.array([0, len(tgt)], dtype=np.int64)
    sb = ctx.scan_batch(tgt, off, src, off)
    T, err, _ = sb.icp(50)
    assert err[0, -1] < 0.05 < err[0, 0]
    assert np.abs(T[0][:3, :3] - T_true[:3, :3]).max() < 1e-3
    assert np.abs(T[0][:3, 3] - T_true[:3, 3]).max() < 0.02
    assert abs(np.linalg.det(T[0][:3, :3]) - 1) < 1e-9
    T_again, err2, _ = sb.icp(2)  # continue from the converged pose: nothing moves
    assert np.abs(T_again[0] - T[0]).max() < 1e-4
    idx, sqd = sb.correspondences()
    assert idx.min() >= 0 and idx.max() < len(tgt)
    # spot-check 2000 correspondences of the last iteration against brute force
    sel = np.random.default_rng(0).choice(len(src), 2000, replace=False)
    # pose used by the last iteration = pose before it; recompute from the first of the two
    sb.set_pose(T[0][None])
    sb.icp(1)
    idx, sqd = sb.correspondences()
    P = O.transform_f32(T[0], src[sel])
    ridx, rsqd = O.knn_brute(tgt, P, 1)
    assert np.array_equal(idx[sel], ridx[:, 0]) and np.array_equal(sqd[sel], rsqd[:, 0])
    sb.close()


@pytest.mark.parametrize("n", [262144, 1048576])
def test_icp_largest_config_sizes(ctx, n):
    """BASELINE configs[3] / [4] sizes (262 144 and 1 048 576 points per scan, 20 iterations), by
    size-independent properties -- convergence to the generating transform, a proper rotation, and exactness
    of a random sample of the final correspondences and of stand-alone k = 3 searches against BRUTE FORCE
    (an oracle that shares no search structure with the product).  The whole-pair comparison with the
    kd-tree oracle at these sizes is test_icp_full_size_matches_oracle."""
    tgt, src, T_true = synth.scan_pair(n, 1)
    off = np.array([0, n], dtype=np.int64)
    sb = ctx.scan_batch(tgt, off, src, off)
    T, err, _ = sb.icp(20)
    assert err[0, -1] < 0.05 < err[0, 0]
    assert np.abs(T[0][:3, :3] - T_true[:3, :3]).max() < 1e-3
    assert np.abs(T[0][:3, 3] - T_true[:3, 3]).max() < 0.02
    assert abs(np.linalg.det(T[0][:3, :3]) - 1) < 1e-9
    sb.set_pose(T[0][None])
    sb.icp(1)
    idx, sqd = sb.correspondences()
    sel = np.random.default_rng(n).choice(n, 1500, replace=False)
    ridx, rsqd = O.knn_brute(tgt, O.transform_f32(T[0], src[sel]), 1)
    assert np.array_equal(idx[sel], ridx[:, 0]) and np.array_equal(sqd[sel], rsqd[:, 0])
    # far from convergence (second iteration from the identity: queries decimetres off the surface, seeded
    # by the first iteration's neighbours; at 1M points this is the ball search)
    sb.set_pose(None)
    T1, _, _ = sb.icp(1)
    sb.set_pose(None)
    sb.icp(2)
    idx, sqd = sb.correspondences()
    ridx, rsqd = O.knn_brute(tgt, O.transform_f32(T1[0], src[sel]), 1)
    assert np.array_equal(idx[sel], ridx[:, 0]) and np.array_equal(sqd[sel], rsqd[:, 0])
    sb.close()
    index = ctx.knn_index(tgt)
    q = src[sel[:500]] + np.float32(0.3)
    gi, gd = index.search(q, 3)
    ri, rd = O.knn_brute(tgt, q, 3)
    assert np.array_equal(gi, ri) and np.array_equal(gd, rd)
    index.close()


@pytest.mark.parametrize("n,iters,pair", [(65536, 50, 0), (262144, 20, 1), (1048576, 20, 1)])
def test_icp_full_size_matches_oracle(ctx, n, iters, pair):
    """BASELINE configs[1] / [3] / [4] sizes against the kd-tree oracle on the WHOLE pair: the run's pose and
    error history within 1e-5 (float32 rounding of the running pose can flip single correspondences between
    two exact implementations that add their float64 sums in different orders), and the correspondences of the
    last iteration -- every point, index and squared distance -- bit-exact against one oracle iteration from
    the pose the product held before it."""
    tgt, src, T_true = synth.scan_pair(n, pair)
    off = np.array([0, n], dtype=np.int64)
    kd = O.KdTree(tgt)
    T_ref, hist = kd.icp_run(src, iters)
    sb = ctx.scan_batch(tgt, off, src, off)
    T, err, _ = sb.icp(iters)
    assert np.abs(T[0] - T_ref).max() < 1e-5
    assert np.abs(err[0] - hist).max() < 1e-5
    assert np.abs(T[0][:3, :3] - T_true[:3, :3]).max() < 1e-3 and np.abs(T[0][:3, 3] - T_true[:3, 3]).max() < 0.02
    idx, sqd = sb.correspondences()
    sb.set_pose(None)
    T_prev, _, _ = sb.icp(iters - 1)  # the pose the last iteration searched with
    T_next, e_ref, idx_ref, sqd_ref = kd.icp_iterate(src, T_prev[0])
    assert np.array_equal(idx, idx_ref) and np.array_equal(sqd, sqd_ref)
    assert np.abs(T[0] - T_next).max() < 1e-9 and abs(err[0, -1] - e_ref) < 1e-10
    sb.close()


@pytest.mark.parametrize("R", [1, 2, 4])
def test_icp_ball_search_is_the_same_search(ctx, monkeypatch, R):
    """Dense clouds take the ball search (GPSCAL_BALL_R, chosen by the level-0 cell size); it must return
    what the fine -> coarse block search returns.  Small cells (0.12 m) put the queries many cells from
    the surface, the regime it is for; both the seedless first iteration and the seeded ones are compared
    with brute force and with the block search."""
    npairs, n = 3, 20000
    tg, to, sr, so, _ = synth.scan_batch(npairs, n)
    out = {}
    for r in (0, R):
        monkeypatch.setenv("GPSCAL_BALL_R", str(r))
        sb = ctx.scan_batch(tg, to, sr, so, cell_size=0.12)
        T1, _, _ = sb.icp(1)
        i1, d1 = sb.correspondences()
        sb.set_pose(None)
        T3, e3, _ = sb.icp(3)
        i3, d3 = sb.correspondences()
        out[r] = (T1.copy(), i1, d1, T3.copy(), i3, d3, e3.copy())
        if r:
            for p in range(npairs):
                a, b = so[p], so[p + 1]
                ridx, rsqd = O.knn_brute(tg[to[p]:to[p + 1]], sr[a:b], 1)  # iteration 1: identity pose
                assert np.array_equal(i1[a:b], ridx[:, 0]) and np.array_equal(d1[a:b], rsqd[:, 0])
            sb.set_pose(None)
            T2, _, _ = sb.icp(2)  # pose the third iteration searched with
            for p in range(npairs):
                a, b = so[p], so[p + 1]
                ridx, rsqd = O.knn_brute(tg[to[p]:to[p + 1]], O.transform_f32(T2[p], sr[a:b]), 1)
                assert np.array_equal(i3[a:b], ridx[:, 0]) and np.array_equal(d3[a:b], rsqd[:, 0])
        sb.close()
    for k in (1, 2, 4, 5):
        assert np.array_equal(out[0][k], out[R][k])
    # two builds group the source points identically (original-index order inside a cell): same sums, bit for bit
    assert np.array_equal(out[0][3], out[R][3]) and np.array_equal(out[0][6], out[R][6])


def _odd_cloud(rng, kind, n):
    p = rng.normal(0, 3, (n, 3))
    if kind == "plane":
        p[:, 2] = 1.5
    elif kind == "line":
        p[:, 1] = -2.0
        p[:, 2] = 0.25
    elif kind == "point":
        p[:] = rng.normal(0, 3, (1, 3))
    elif kind == "outlier":
        p[0] = [1.0e4, -2.0e4, 3.0e3]
    elif kind == "offset":
        p += [2.0e5, 1.0e5, 50.0]
    elif kind == "dupes":
        p[n // 2:] = p[: n - n // 2]
    elif kind == "lattice":
        p = np.round(p * 2) / 2  # many exact ties in distance
    elif kind == "nans":
        p[::9, rng.integers(0, 3)] = np.nan  # target points that are never indexed nor returned
    return p.astype(np.float32)


def _brute_finite(t, q, k):
    """Brute force over the finite target points only (non-finite ones are never indexed nor returned by the
    product; the oracle's brute force is not defined for them), indices mapped back to the full cloud."""
    fin = np.flatnonzero(np.isfinite(t).all(axis=1))
    ri, rd = O.knn_brute(t[fin], q, k)
    return np.where(ri >= 0, fin[np.maximum(ri, 0)], -1).astype(np.int32), rd


@pytest.mark.parametrize("ball", [None, "2"])
def test_knn_and_icp_odd_geometry_vs_brute_force(ctx, monkeypatch, ball):
    """Degenerate bounding boxes (planes, lines, one point), far outliers, large offsets, duplicates and a
    lattice full of distance ties, at small random sizes: k-NN and two ICP iterations against brute force."""
    if ball:
        monkeypatch.setenv("GPSCAL_BALL_R", ball)
    rng = np.random.default_rng(2024)
    kinds = ["plane", "line", "point", "outlier", "offset", "dupes", "lattice", "blob", "nans"]
    src_kinds = [k for k in kinds if k != "nans"]  # a source must be finite
    tgts, srcs = [], []
    for kind in kinds:
        m, n = int(rng.integers(1, 900)), int(rng.integers(1, 700))
        t = _odd_cloud(rng, kind, m)
        s = _odd_cloud(rng, rng.choice(src_kinds), n) if kind != "offset" else (t[rng.integers(0, m, n)] + rng.normal(0, 0.3, (n, 3))).astype(np.float32)
        tgts.append(t)
        srcs.append(s)
        ix = ctx.knn_index(t)
        for k in (1, 3):
            gi, gd = ix.search(s, k)
            ri, rd = _brute_finite(t, s, k)
            assert np.array_equal(gi, ri) and np.array_equal(gd, rd), (kind, k)
        ix.close()
    tg, sr = np.concatenate(tgts), np.concatenate(srcs)
    to = np.cumsum([0] + [len(t) for t in tgts]).astype(np.int64)
    so = np.cumsum([0] + [len(t) for t in srcs]).astype(np.int64)
    sb = ctx.scan_batch(tg, to, sr, so)
    T1, _, _ = sb.icp(1)
    i1, d1 = sb.correspondences()
    sb.set_pose(None)
    sb.icp(2)
    i2, d2 = sb.correspondences()
    for p, kind in enumerate(kinds):
        a, b = so[p], so[p + 1]
        ri, rd = _brute_finite(tgts[p], srcs[p], 1)
        assert np.array_equal(i1[a:b], ri[:, 0]) and np.array_equal(d1[a:b], rd[:, 0]), kind
        assert np.isfinite(T1[p]).all(), kind
        ri, rd = _brute_finite(tgts[p], O.transform_f32(T1[p], srcs[p]), 1)
        assert np.array_equal(i2[a:b], ri[:, 0]) and np.array_equal(d2[a:b], rd[:, 0]), kind
    sb.close()


def test_knn_volume_filling_cloud_with_millions_of_cells(ctx):
    """Not lidar-like, but the ABI must survive it: 200 000 points filling a 500 m cube (level 0 gets ~18 M
    cells, most of them empty; the scan over the cell counters runs three levels deep), explicit tiny cells
    that hit the 2^25-cell cap, and strided / device-resident inputs."""
    rng = np.random.default_rng(8)
    t = rng.uniform(-250, 250, (200000, 3)).astype(np.float32)
    q = rng.uniform(-260, 260, (300, 3)).astype(np.float32)
    ri, rd = O.knn_brute(t, q, 2)
    for cell in (0.0, 0.05):
        ix = ctx.knn_index(t, cell_size=cell)
        gi, gd = ix.search(q, 2)
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd), cell
        ix.close()
    import torch
    t32 = np.zeros((len(t), 8), dtype=np.float32)  # PointXYZI-like 32-byte records
    t32[:, :3] = t
    ix = ctx.knn_index(torch.from_numpy(t32).cuda(), stride_bytes=32)
    gi, gd = ix.search(torch.from_numpy(q).cuda(), 2)
    assert np.array_equal(gi, ri) and np.array_equal(gd, rd)
    T, hist = ix.icp_run(q, 3)
    assert np.isfinite(T).all()
    ix.close()


def test_icp_small_source_clouds_match_oracle(ctx):
    """Sources of 1 .. 1000 points (the source grouping of a small cloud is one 8x8 tile per layer, counted in
    LDS: a path the large benchmarks never take) against the oracle's ICP."""
    tgt, src, _ = synth.scan_pair(3000, 4)
    sizes = [1, 5, 33, 64, 65, 257, 1000]
    tg = np.concatenate([tgt] * len(sizes)).astype(np.float32)
    to = (np.arange(len(sizes) + 1) * len(tgt)).astype(np.int64)
    srcs = [src[7 * k:7 * k + n] for k, n in enumerate(sizes)]
    sr = np.concatenate(srcs).astype(np.float32)
    so = np.cumsum([0] + sizes).astype(np.int64)
    sb = ctx.scan_batch(tg, to, sr, so)
    T, err, _ = sb.icp(3)
    idx, sqd = sb.correspondences()
    kd = O.KdTree(tgt)
    for p, n in enumerate(sizes):
        if n >= 3:
            T_ref, hist = kd.icp_run(srcs[p], 3)
            assert np.abs(T[p] - T_ref).max() < 1e-6 and np.abs(err[p] - hist).max() < 1e-6, n
    # correspondences of the third iteration of the largest cloud against brute force
    sb.set_pose(None)
    T2, _, _ = sb.icp(2)
    ridx, rsqd = O.knn_brute(tgt, O.transform_f32(T2[-1], srcs[-1]), 1)
    assert np.array_equal(idx[so[-2]:], ridx[:, 0]) and np.array_equal(sqd[so[-2]:], rsqd[:, 0])
    sb.close()


def test_icp_ball_search_on_tiny_and_degenerate_targets(ctx, monkeypatch):
    """Forced ball search where the level ladder is degenerate: targets of 1, 2 and 7 points, a target of
    coincident points, a source far outside the target's box."""
    monkeypatch.setenv("GPSCAL_BALL_R", "3")
    rng = np.random.default_rng(11)
    tgts = [rng.normal(0, 1, (1, 3)), rng.normal(0, 1, (2, 3)), rng.normal(0, 1, (7, 3)),
            np.repeat(rng.normal(0, 1, (1, 3)), 50, axis=0), rng.normal(0, 0.5, (300, 3))]
    srcs = [rng.normal(0, 2, (40, 3)), rng.normal(0, 2, (40, 3)), rng.normal(0, 2, (64, 3)),
            rng.normal(0, 2, (70, 3)), rng.normal(0, 0.5, (200, 3)) + 40.0]
    tg = np.concatenate(tgts).astype(np.float32)
    sr = np.concatenate(srcs).astype(np.float32)
    to = np.cumsum([0] + [len(t) for t in tgts]).astype(np.int64)
    so = np.cumsum([0] + [len(t) for t in srcs]).astype(np.int64)
    sb = ctx.scan_batch(tg, to, sr, so)
    for iters in (1, 2):  # without and with remembered neighbours
        sb.set_pose(None)
        Tprev = np.tile(np.eye(4), (len(tgts), 1, 1))
        if iters == 2:
            Tprev, _, _ = sb.icp(1)
            sb.set_pose(None)
        sb.icp(iters)
        idx, sqd = sb.correspondences()
        for p in range(len(tgts)):
            a, b = so[p], so[p + 1]
            ridx, rsqd = O.knn_brute(tg[to[p]:to[p + 1]], O.transform_f32(Tprev[p], sr[a:b]), 1)
            assert np.array_equal(idx[a:b], ridx[:, 0]) and np.array_equal(sqd[a:b], rsqd[:, 0]), (iters, p)
    sb.close()


def test_icp_two_builds_are_bit_identical(ctx):
    """The counting sort that groups the source points takes slots with atomics; order_runs_kernel then puts
    every cell into original-index order, so two builds add the float64 sums in the same order.  The cloud has a
    cell of ~300 points (ordered by a whole wave) and one of ~700 (left as the atomics put it: only the
    correspondences, not the last bits of the pose, are compared for that one)."""
    tg, to, sr, so, _ = synth.scan_batch(2, 16384)
    rng = np.random.default_rng(5)
    dense = sr[:300] * 0 + sr[100] + rng.normal(0, 0.004, (300, 3)).astype(np.float32)
    sr2 = np.concatenate([sr[:so[1]], dense, sr[so[1]:]]).astype(np.float32)
    so2 = np.array([0, so[1] + 300, so[2] + 300], dtype=np.int64)
    runs = []
    for _ in range(3):
        sb = ctx.scan_batch(tg, to, sr2, so2)
        T, err, _ = sb.icp(8)
        idx, sqd = sb.correspondences()
        runs.append((T.copy(), err.copy(), idx, sqd))
        sb.close()
    for r in runs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(runs[0], r))
    blob = sr[:700] * 0 + sr[200] + rng.normal(0, 0.002, (700, 3)).astype(np.float32)
    sr3 = np.concatenate([sr, blob]).astype(np.float32)
    so3 = np.array([0, so[1], so[2] + 700], dtype=np.int64)
    a = ctx.scan_batch(tg, to, sr3, so3)
    b = ctx.scan_batch(tg, to, sr3, so3)
    Ta, _, _ = a.icp(4)
    Tb, _, _ = b.icp(4)
    assert np.abs(Ta - Tb).max() < 1e-12
    assert all(np.array_equal(x, y) for x, y in zip(a.correspondences(), b.correspondences()))
    a.close()
    b.close()


def test_icp_run_in_two_calls_equals_one_call(ctx):
    """A run may be continued: icp(8) followed by icp(12) is the 20-iteration run, bit for bit (poses and final
    correspondences), and alternating between iteration counts keeps one captured graph per count -- the third and
    fourth calls below replay the graphs of the first two."""
    tg, to, sr, so, _ = synth.scan_batch(9, 8192)
    one = ctx.scan_batch(tg, to, sr, so)
    T20, _, _ = one.icp(20)
    c20 = one.correspondences()
    two = ctx.scan_batch(tg, to, sr, so)
    two.icp(8)
    Tb, _, _ = two.icp(12)
    assert np.array_equal(T20, Tb)
    assert all(np.array_equal(a, b) for a, b in zip(c20, two.correspondences()))
    two.set_pose(None)
    two.icp(8)
    Tc, _, _ = two.icp(12)
    assert np.array_equal(T20, Tc)
    one.close()
    two.close()


def _ragged_batch(npairs, n):
    """Pairs of different sizes, a tiny source and a tiny target among them, stored back to back."""
    tg, to, sr, so, _ = synth.scan_batch(npairs, n)
    rng = np.random.default_rng(77)
    tgts, srcs = [], []
    for p in range(npairs):
        t, s = tg[to[p]:to[p + 1]], sr[so[p]:so[p + 1]]
        if p == 3:
            s = s[:5]            # a source of five points
        elif p == 5:
            t = t[::9]           # a sparse target
        elif p == 7:
            s = s[:1]
        else:
            s = s[: int(rng.integers(n // 3, n))]
            t = t[: int(rng.integers(n // 2, n))]
        tgts.append(t)
        srcs.append(s)
    to2 = np.cumsum([0] + [len(t) for t in tgts]).astype(np.int64)
    so2 = np.cumsum([0] + [len(s) for s in srcs]).astype(np.int64)
    return np.concatenate(tgts), to2, np.concatenate(srcs), so2


@pytest.mark.parametrize("ragged", [False, True])
def test_icp_graph_chains_and_slices_agree_with_each_other_and_the_oracle(ctx, monkeypatch, ragged):
    """What the headline benchmark runs -- the captured graph with several step -> solve chains, chain-local block
    and partial-sum offsets, workgroup slices by arithmetic -- against the whole-batch launches of the profiling mode,
    against one chain, against the table-driven slices, against the other row walk of the grid search (per-lane lists /
    row by row: the same exact search, so the same bits), against the one-launch persistent kernel (GPSCAL_ICP_PERSISTENT=1:
    all iterations of a small batch in one launch, bit-identical to the 256-thread graph path), and across the three workgroup sizes of the step kernel (last bits of the pose, never a correspondence); and against the kd-tree oracle.  12 pairs: two chains by default, four forced; the ragged batch has a
    five-point and a one-point source and a sparse target, so chains split unevenly and slices come from the table."""
    npairs, n, iters = 12, 4096, 12
    if ragged:
        tg, to, sr, so = _ragged_batch(npairs, n)
    else:
        tg, to, sr, so, _ = synth.scan_batch(npairs, n)
    variants = {"default": {}, "persistent": {"GPSCAL_ICP_PERSISTENT": "1"}, "chains1": {"GPSCAL_ICP_CHAINS": "1"},
                "chains4": {"GPSCAL_ICP_CHAINS": "4"}, "table": {"GPSCAL_ICP_UNIFORM": "0"},
                "wg128": {"GPSCAL_STEP_BLOCK": "128", "GPSCAL_ICP_CHAINS": "4"}, "wg256": {"GPSCAL_STEP_BLOCK": "256"},
                "rowwalk": {"GPSCAL_STEP_FLAT": "0"}}  # (a batch this small walks the rows of a level as per-lane lists by default)
    runs = {}
    for name, env in variants.items():
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sb = ctx.scan_batch(tg, to, sr, so)
        for k in env:
            monkeypatch.delenv(k)
        T, err, _ = sb.icp(iters)                 # the replayed graph
        idx, sqd = sb.correspondences()
        sb.set_pose(None)
        Tp, errp, _ = sb.icp(iters, profile=True)  # whole-batch launches, one stream
        assert np.array_equal(T, Tp) and np.array_equal(err, errp), name
        runs[name] = (T.copy(), err.copy(), idx, sqd)
        if name == "default":
            sb.set_pose(None)
            T_prev, _, _ = sb.icp(iters - 1)
        sb.close()
    for name in ("chains1", "chains4", "table", "rowwalk"):
        assert all(np.array_equal(a, b) for a, b in zip(runs["default"], runs[name])), name
    # (the persistent kernel is built for 256-thread workgroups; a batch this small takes 512 by default)
    assert all(np.array_equal(a, b) for a, b in zip(runs["wg256"], runs["persistent"]))
    # another workgroup size adds the float64 sums in another order: last bits of the pose, never a correspondence
    for name in ("wg128", "wg256"):
        assert np.abs(runs[name][0] - runs["default"][0]).max() < 1e-9, name
        assert np.array_equal(runs[name][2], runs["default"][2]) and np.array_equal(runs[name][3], runs["default"][3]), name
    T, err, idx, sqd = runs["default"]
    for p in range(npairs):
        t, s = tg[to[p]:to[p + 1]], sr[so[p]:so[p + 1]]
        kd = O.KdTree(t)
        if len(s) >= 3:
            T_ref, hist = kd.icp_run(s, iters)
            assert np.abs(T[p] - T_ref).max() < 1e-5 and np.abs(err[p] - hist).max() < 1e-5, p
        ridx, rsqd = O.knn_brute(t, O.transform_f32(T_prev[p], s), 1)
        assert np.array_equal(idx[so[p]:so[p + 1]], ridx[:, 0]) and np.array_equal(sqd[so[p]:so[p + 1]], rsqd[:, 0]), p


def test_icp_runs_of_a_large_batch_repeat_bit_for_bit(ctx):
    """A batch of a million source points (256-thread workgroups, two chains in the graph): every run from the same
    pose gives the same bits -- the first run, later replays of the graph and the profiling mode's whole-batch
    launches -- and one pair is checked against the kd-tree oracle.  (Until the step kernel's sums were rewritten for
    instruction count, large batches switched to a second kernel in the converged iterations of later runs, which
    changed the last bits between the first run and the later ones; that kernel is gone.)"""
    npairs, n, iters = 16, 65536, 30
    tg, to, sr, so, _ = synth.scan_batch(npairs, n)
    sb = ctx.scan_batch(tg, to, sr, so)
    runs = []
    for k in range(4):
        sb.set_pose(None)
        T, err, _ = sb.icp(iters, profile=(k == 3))
        idx, sqd = sb.correspondences()
        runs.append((T.copy(), err.copy(), idx, sqd))
    sb.close()
    for r in runs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(runs[0], r))
    T_ref, hist = O.KdTree(tg[to[5]:to[6]]).icp_run(sr[so[5]:so[6]], iters)
    assert np.abs(runs[1][0][5] - T_ref).max() < 1e-5 and np.abs(runs[1][1][5] - hist).max() < 1e-5


@pytest.mark.parametrize("npairs,n,iters", [(8, 262144, 20), (4, 1048576, 10)])
def test_icp_batches_at_the_benchmark_shapes_match_oracle(ctx, npairs, n, iters):
    """Many-pair batches at the sizes of BASELINE configs[3] / [4] through ONE scan batch (the index of such a batch
    is gigabytes, several chains run in the graph): every pair's pose and error history against the kd-tree oracle,
    and a sample of every pair's last-iteration correspondences bit-exact against brute force."""
    tg, to, sr, so, T_true = synth.scan_batch(npairs, n)
    sb = ctx.scan_batch(tg, to, sr, so)
    T, err, _ = sb.icp(iters)
    idx, sqd = sb.correspondences()
    sb.set_pose(None)
    T_prev, _, _ = sb.icp(iters - 1)
    sb.close()
    rng = np.random.default_rng(n)
    for p in range(npairs):
        t, s = tg[to[p]:to[p + 1]], sr[so[p]:so[p + 1]]
        T_ref, hist = O.KdTree(t).icp_run(s, iters)
        assert np.abs(T[p] - T_ref).max() < 1e-5 and np.abs(err[p] - hist).max() < 1e-5, p
        sel = rng.choice(n, 600, replace=False)
        ridx, rsqd = O.knn_brute(t, O.transform_f32(T_prev[p], s[sel]), 1)
        assert np.array_equal(idx[so[p] + sel], ridx[:, 0]) and np.array_equal(sqd[so[p] + sel], rsqd[:, 0]), p


# -------------------------------------------------------------------- track
def _segments(nseg, poses, seed, dropout=0.0):
    d = synth.track_segments(nseg, poses, seed=seed, dropout=dropout)
    lat, lon, gt, valid = d["gps"]
    lat = np.where(valid, lat, 90.0)
    lon = np.where(valid, lon, 180.0)
    la, lo, _ = O.gap_fill(lat, lon, gt)
    return d, la, lo, gt


def test_weights_match_oracle(ctx):
    d, la, lo, gt = _segments(1, 500, 2)
    slam = d["slam"]
    enu = O.gps_to_enu(la, lo, gt, slam)
    w = ctx.weights_speed(slam)
    assert np.array_equal(w, O.weights_speed(slam))  # same operations, bit-exact
    _, _, cal, _ = O.track_fit(slam, enu, w)
    wi = ctx.weights_irls(slam, enu, cal)
    np.testing.assert_allclose(wi, O.weights_irls(slam, enu, cal), rtol=1e-14)


def test_weights_known_answer(ctx):
    # weight_calculation.cc:4-27, 30-78 evaluated by hand (the same case as test_oracle_cpu.test_irls_weights_known_answer):
    # speed weights 1, 1.1 / 2.2, min(4.4 / 2.2, 1), 1; residuals 0.5, 0.005 (clamped to 0.01), 2, 0.25
    slam = np.array([[0, 0, 10, 0], [0.55, 0, 10, 1], [0.55, 1.1, 10, 2], [0.55, 5.5, 10, 3]], dtype=np.float64)
    fit = np.array([[5, 7, 10, 0], [6, 7, 10, 1], [6, 8, 10, 2], [6, 12, 10, 3]], dtype=np.float64)
    enu = fit.copy()
    enu[:, 0] += [0.5, 0.005, 2.0, 0.25]
    assert ctx.weights_speed(slam).tolist() == [1.0, 0.5, 1.0, 1.0]
    np.testing.assert_allclose(ctx.weights_irls(slam, enu, fit), [2.0, 50.0, 0.5, 4.0], rtol=1e-15)


def test_gps_to_enu_and_back_match_oracle(ctx, gps_log_bytes):
    st = 1494650700.0 + np.arange(1000)
    slam = np.zeros((1000, 4))
    slam[:, 2] = 10
    slam[:, 3] = st
    lat, lon, t = O.parse_gprmc(gps_log_bytes, st[0], st[-1])
    enu = ctx.gps_to_enu(lat, lon, t, slam)
    ref = O.gps_to_enu(lat, lon, t, slam)
    assert enu.shape == ref.shape
    assert np.abs(enu[:, :2] - ref[:, :2]).max() < 1e-6
    assert np.array_equal(enu[:, 2:], ref[:, 2:])
    # the reference-probe KAT through the GPU path
    assert abs(enu[0, 0] - 3450164.856218) < 1e-5 and abs(enu[0, 1] - 400633250.787481) < 1e-5
    e5 = np.c_[ref, np.ones(len(ref))]
    ll, alt = ctx.enu_to_wgs(e5)
    rll, ralt = O.local_to_wgs(e5)
    assert np.abs(ll - rll).max() < 1e-10 and np.array_equal(alt, ralt)
    assert abs(ll[0, 0] - 121.398330784171) < 1e-10 and abs(ll[0, 1] - 31.177944836485) < 1e-10
    for method in ("UTM", "Gaussion"):
        for band in (3, 6):
            xy = ctx.wgs_to_enu(lat, lon, method, band)
            rxy = O.wgs_to_local(lat, lon, 0 if method == "UTM" else 1, band)
            assert np.abs(xy - rxy).max() < 1e-6, (method, band)


def test_gps_to_enu_drops_stamps_after_last_fix(ctx):
    d, la, lo, gt = _segments(1, 100, 4)
    slam = d["slam"].copy()
    slam[-5:, 3] = gt[-1] + 1.0 + np.arange(5)
    enu = ctx.gps_to_enu(la, lo, gt, slam)
    ref = O.gps_to_enu(la, lo, gt, slam)
    assert len(enu) == len(ref) == 95


@pytest.mark.parametrize("n", [2, 3, 64, 257, 400, 1250, 3000])
def test_track_fit_matches_oracle(ctx, n):
    d, la, lo, gt = _segments(1, n, 10 + n)
    slam = d["slam"]
    enu = O.gps_to_enu(la, lo, gt, slam)
    w = O.weights_speed(slam)
    T, rot, cal = ctx.track_fit(slam, enu, w)
    T_ref, rot_ref, cal_ref, _ = O.track_fit(slam, enu, w)
    if n > 2:  # two points are collinear: det H2 = 0, rotation and reflection fit equally well
        assert np.abs(T - T_ref).max() < 1e-8
    assert np.abs(rot - rot_ref).max() < 1e-7
    assert np.abs(cal[:, :2] - cal_ref[:, :2]).max() < 1e-6
    assert np.array_equal(cal[:, 2:], cal_ref[:, 2:])


def test_track_fit_reflection_case_matches_oracle(ctx):
    d, la, lo, gt = _segments(1, 300, 77)
    slam = d["slam"].copy()
    slam[:, 1] *= -1  # mirrored: det H2 < 0
    enu = O.gps_to_enu(la, lo, gt, slam)
    w = np.ones(len(slam))
    T, rot, cal = ctx.track_fit(slam, enu, w)
    T_ref, rot_ref, cal_ref, _ = O.track_fit(slam, enu, w)
    assert T_ref[2, 2] == -1.0 and T[2, 2] == -1.0 and T[2, 3] == 2.0
    assert np.abs(T - T_ref).max() < 1e-8
    assert np.abs(cal[:, :2] - cal_ref[:, :2]).max() < 1e-6


def test_track_fit_size_mismatch_is_an_error_not_exit(ctx):
    from gpscalibration_amd import GpscalError
    with pytest.raises(GpscalError):
        ctx.track_fit(np.zeros((10, 4)), np.zeros((9, 4)), np.ones(10))


def test_long_segment_batched_matches_oracle(ctx):
    # ragged segments, 30 % dropout bursts (BASELINE configs[4] flavour)
    nseg = 12
    d, la, lo, gt = _segments(nseg, 400, 5, dropout=0.3)
    slam = d["slam"]
    enu = O.gps_to_enu(la, lo, gt, slam)
    cuts = np.r_[0, np.cumsum([400, 380, 420, 10, 790, 400, 3, 397, 400, 400, 400, 800])].astype(np.int32)
    assert cuts[-1] == len(slam)
    w, fit = ctx.long_segment(slam, enu, 5, seg_offsets=cuts)
    for s in range(nseg):
        a, b = cuts[s], cuts[s + 1]
        w_ref, fit_ref = O.long_segment(slam[a:b], enu[a:b], 5)
        np.testing.assert_allclose(w[a:b], w_ref, rtol=1e-6, err_msg=str(s))
        assert np.abs(fit[a:b, :2] - fit_ref[:, :2]).max() < 1e-6, s
        assert np.array_equal(fit[a:b, 2:], fit_ref[:, 2:])


def test_track_pipeline_kml_degrees(ctx):
    """Long pass weights -> short pass fits -> merge -> WGS84: KML degrees vs the oracle chain."""
    d, la, lo, gt = _segments(4, 300, 21)
    slam, so = d["slam"], d["seg_off"]
    enu = ctx.gps_to_enu(la, lo, gt, slam)
    enu_ref = O.gps_to_enu(la, lo, gt, slam)
    w, _ = ctx.long_segment(slam, enu, 5, seg_offsets=so)
    T, rot, cal = ctx.track_fit(slam, enu, w, seg_offsets=so)
    acc = acc_ref = None
    for s in range(4):
        a, b = so[s], so[s + 1]
        w_ref, _ = O.long_segment(slam[a:b], enu_ref[a:b], 5)
        _, _, cal_ref, _ = O.track_fit(slam[a:b], enu_ref[a:b], w_ref)
        acc = O.merge_short(acc, cal[a:b], w[a:b])
        acc_ref = O.merge_short(acc_ref, cal_ref, w_ref)
    ll, _ = ctx.enu_to_wgs(acc)
    ll_ref, _ = O.local_to_wgs(acc_ref)
    assert np.abs(ll - ll_ref).max() < 1e-9  # degrees (north_star bar: 1e-6)


def test_height_compensate_matches_oracle(ctx):
    rng = np.random.default_rng(4)
    p = np.c_[np.cumsum(rng.normal(0, 1, size=(200, 3)), axis=0), 10.0 + np.arange(200)]
    out = ctx.height_compensate(p)
    np.testing.assert_allclose(out, O.height_compensate(p), rtol=1e-13, atol=1e-12)


def test_rccl_world_of_one(ctx):
    """The RCCL path with a single rank: exercises dlopen, communicator setup and both the
    equal-count and ragged code paths (multi-rank runs are the driver's 8-GPU bench)."""
    import ctypes as C
    L = ctx._L
    uid = C.create_string_buffer(128)
    assert L.gpscal_comm_unique_id(uid) == 0
    rc = L.gpscal_comm_init(ctx._h, uid, 0, 1)
    assert rc == 0, L.gpscal_last_error(ctx._h)
    local = np.arange(32, dtype=np.float64)
    counts = np.array([32], dtype=np.int32)
    out = np.zeros(32)
    assert L.gpscal_allgather_chains(ctx._h, local.ctypes.data, counts.ctypes.data, out.ctypes.data) == 0
    assert np.array_equal(out, local)
    assert L.gpscal_comm_destroy(ctx._h) == 0


def test_gcj_bd_transforms_match_oracle(ctx):
    """gpscal_gps_to_gcj / gcj_to_bd / bd_to_gcj (gps_process.cc:526-595, 1127-1207) vs the oracle:
    float64 polynomials and sines; device libm vs glibc -> 1e-12 degrees."""
    rng = np.random.default_rng(3)
    ll = np.c_[rng.uniform(60, 150, 5000), rng.uniform(-10, 65, 5000)]  # in and out of the China box
    gcj = ctx.mars(ll, "gps_to_gcj")
    assert np.abs(gcj - O.mars(ll, "gps_to_gcj")).max() < 1e-12
    out = (ll[:, 0] < 72.004) | (ll[:, 0] > 137.8347) | (ll[:, 1] < 0.8293) | (ll[:, 1] > 55.8271)
    assert out.any() and np.array_equal(gcj[out], ll[out])
    bd = ctx.mars(gcj, "gcj_to_bd")
    assert np.abs(bd - O.mars(gcj, "gcj_to_bd")).max() < 1e-12
    assert np.abs(ctx.mars(bd, "bd_to_gcj") - O.mars(bd, "bd_to_gcj")).max() < 1e-12
