"""GPU parity of the LOAM odometry loop (laserOdometry.cpp:585-1029) against the oracle on
synthetic VLP-16-like feature sweeps.  Parity bar: the two sides share the float32 per-point
formulas but not libm (device sinf/cosf vs glibc) nor the summation order of the normal
equations, and a flipped ring-neighbour changes a residual, so transforms agree to 2e-4
(rad / m) after up to 25 iterations; this part of the reference is parity-unpinned
(PCL/FLANN, OpenCV absent, no fixtures)."""
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


@pytest.fixture(scope="module")
def sweeps():
    A = synth.loam_sweep((0, 0, 0, 0, 0, 0), seed=1)
    motions = [(0.004, 0.015, -0.003, 0.05, 0.01, 0.45), (0.0, -0.02, 0.0, -0.1, 0.0, 0.3), (0, 0, 0, 0, 0, 0.02)]
    return A, [synth.loam_sweep(m, seed=10 + k) for k, m in enumerate(motions)]


def test_loam_transforms_match_oracle(ctx, sweeps):
    A, Bs = sweeps
    tr = np.array([0.01, -0.02, 0.005, 0.1, -0.05, 0.4], dtype=np.float32)
    pts = Bs[0]["less_flat"][:2000]
    for to_end in (False, True):
        got = ctx.loam_transform(tr, pts, to_end)
        ref = O.lo_transform(tr, pts, to_end)
        assert np.abs(got - ref).max() < 2e-5


def test_loam_odometry_batched_matches_oracle(ctx, sweeps):
    A, Bs = sweeps
    n = len(Bs)
    tin = np.zeros((n, 6), dtype=np.float32)
    tin[1] = [0, -0.01, 0, -0.05, 0, 0.15]  # a warm start, as the previous sweep's result would be
    ssum = np.tile(np.array([0.01, 0.3, -0.02, 1.0, 0.2, 5.0], dtype=np.float32), (n, 1))
    tr, iters, nsel, sout = ctx.loam_odometry([b["sharp"] for b in Bs], [b["flat"] for b in Bs],
                                              [A["less_sharp"]] * n, [A["less_flat"]] * n, tin, ssum)
    for k in range(n):
        r_tr, r_it, r_ns = O.lo_match(Bs[k]["sharp"], Bs[k]["flat"], A["less_sharp"], A["less_flat"], tin[k])
        assert iters[k] == r_it, k
        assert abs(int(nsel[k]) - r_ns) <= max(3, r_ns // 200), (k, nsel[k], r_ns)
        assert np.abs(tr[k] - r_tr).max() < 2e-4, (k, tr[k], r_tr)
        assert np.abs(sout[k] - O.lo_accumulate(ssum[k], r_tr)).max() < 5e-4
    # the loop moves the estimate towards the true motion: forward motion of 0.45 m shows up as tz < 0
    assert tr[0][5] < -0.1 and tr[2][5] > -0.05


def test_loam_odometry_too_few_last_points_is_a_noop(ctx, sweeps):
    A, Bs = sweeps
    tin = np.array([[0.01, 0.02, 0.03, 0.1, 0.2, 0.3]], dtype=np.float32)
    tr, iters, nsel, _ = ctx.loam_odometry([Bs[0]["sharp"]], [Bs[0]["flat"]], [A["less_sharp"][:5]], [A["less_flat"][:50]], tin)
    assert iters[0] == 0 and np.array_equal(tr[0], tin[0])  # laserOdometry.cpp:569


def test_loam_mapping_batched_matches_oracle(ctx, sweeps):
    """laserMapping.cpp:748-1018: the map is a denser sweep of the same scene, the stacks are the
    features of three other sweeps, each started from a perturbed transformTobeMapped."""
    A, Bs = sweeps
    M = synth.loam_sweep((0, 0, 0, 0, 0, 0), seed=3, n_az=1800)
    n = len(Bs)
    tin = np.array([[0.003, -0.01, 0.002, 0.08, -0.02, 0.12], [0, 0, 0, 0, 0, 0], [-0.004, 0.006, 0.0, -0.05, 0.03, -0.1]],
                   dtype=np.float32)
    tr, iters, nsel = ctx.loam_mapping([b["less_sharp"] for b in Bs], [b["flat"] for b in Bs],
                                       [M["less_sharp"]] * n, [M["less_flat"]] * n, tin)
    for k in range(n):
        r_tr, r_it, r_ns = O.lm_match(Bs[k]["less_sharp"], Bs[k]["flat"], M["less_sharp"], M["less_flat"], tin[k])
        assert iters[k] == r_it, (k, iters[k], r_it)
        assert abs(int(nsel[k]) - r_ns) <= max(3, r_ns // 200), (k, nsel[k], r_ns)
        assert np.abs(tr[k] - r_tr).max() < 2e-4, (k, tr[k], r_tr)
        assert r_ns >= 50 and r_it >= 2


def test_loam_mapping_small_map_is_a_noop(ctx, sweeps):
    A, Bs = sweeps
    tin = np.array([[0.01, 0.02, 0.03, 0.1, 0.2, 0.3]], dtype=np.float32)
    tr, iters, nsel = ctx.loam_mapping([Bs[0]["less_sharp"]], [Bs[0]["flat"]], [A["less_sharp"][:10]], [A["less_flat"]], tin)
    assert iters[0] == 0 and np.array_equal(tr[0], tin[0])  # laserMapping.cpp:748
