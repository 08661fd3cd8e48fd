"""GPU parity of scanRegistration's feature extraction (scanRegistration.cpp:238-674) and of the
VoxelGrid filter against the oracle, on synthetic 16-ring sweeps of a street scene.

Parity bar: ring assignment, ring order, curvature, rejection flags, the per-sector sort and the
picking are integer / exact-float32 work on identical inputs, so point SELECTION and xyz are
bit-exact.  The intensity channel carries relTime, which goes through atan2 (device libm vs
glibc, both evaluated in float64 and rounded to float32): tolerance 2e-6.  sqrtf and float division
are IEEE-exact on both sides (the device build uses no fast-math).  Parity unpinned
against the reference itself (PCL absent, no fixtures): see oracle/sr_oracle.c."""
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
def raw():
    W = synth.lidar_world(0)
    return [synth.raw_sweep(W, pos=(10, 0.5), yaw=0.05, vel=(8, 0), seed=1, nan_every=997),
            synth.raw_sweep(W, pos=(55, -1.0), yaw=-0.3, vel=(5, 1), yaw_rate=0.4, seed=2),
            synth.raw_sweep(W, pos=(120, 0.0), yaw=3.0, seed=3, n_az=900),
            synth.raw_sweep(W, pos=(200, 2.0), yaw=1.2, seed=4, n_az=3600)]  # 57.6k points, 600-point sectors


def _same_cloud(a, b, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a[:, :3], b[:, :3]), what
    assert np.abs(a[:, 3] - b[:, 3]).max(initial=0) <= 2e-6, what


def test_scan_registration_matches_oracle(ctx, raw):
    got = ctx.scan_registration(raw)
    for b, sweep in enumerate(raw):
        ref = O.sr_extract(sweep)
        for name in ("full", "sharp", "less_sharp", "flat"):
            _same_cloud(got[b][name], ref[name], (b, name))
        # less_flat is a centroid of up to a few points per voxel: same cells, float32 sums in the same order
        _same_cloud(got[b]["less_flat"], ref["less_flat"], (b, "less_flat"))
        assert len(ref["sharp"]) > 50 and len(ref["flat"]) > 1000


def test_scan_registration_missing_ring_and_tiny_inputs(ctx, raw):
    """A sweep without ring 4 exercises the reference's span bookkeeping quirk (SR:480-487: the ring
    before the hole loses its span, the hole's index re-scans earlier rings); tiny and all-NaN sweeps
    must come back empty rather than fault."""
    s = raw[0]
    ang = np.degrees(np.arctan2(s[:, 2], np.hypot(s[:, 0], s[:, 1])))
    holed = s[~(np.abs(ang + 7) < 0.5)]
    tiny = s[:7].copy()
    nans = np.full((5, 3), np.nan, dtype=np.float32)
    got = ctx.scan_registration([holed, tiny, nans], less_flat_factor=6)
    ref = O.sr_extract(holed)
    assert len(ref["full"]) < len(O.sr_extract(s)["full"])
    for name in ("full", "sharp", "less_sharp", "flat", "less_flat"):
        _same_cloud(got[0][name], ref[name], name)
    rt = O.sr_extract(tiny)
    for name in ("full", "sharp", "less_sharp", "flat", "less_flat"):
        _same_cloud(got[1][name], rt[name], ("tiny", name))
        assert len(got[2][name]) == 0


def test_scan_registration_maximum_sweeps_in_few_rings(ctx):
    """59 000 points on one ring and on two rings (sectors of ~10 000 points: far beyond the LDS sort
    window), and a structureless 60 000-point cloud: same feature clouds as the oracle."""
    a = np.linspace(0, 2 * np.pi, 59000, endpoint=False)
    one = np.stack([10 * np.cos(a) + 0.2 * np.sin(7 * a), 10 * np.sin(a), np.zeros_like(a)], 1).astype(np.float32)
    two = one.copy()
    two[::2, 2] = np.float32(10 * np.tan(np.deg2rad(7.0)))
    rnd = np.random.default_rng(9).normal(0, 20, (60000, 3)).astype(np.float32)
    sweeps = [one, two, rnd]
    got = ctx.scan_registration(sweeps)
    for b, sweep in enumerate(sweeps):
        ref = O.sr_extract(sweep)
        for name in ("full", "sharp", "less_sharp", "flat", "less_flat"):
            _same_cloud(got[b][name], ref[name], (b, name))


def test_voxel_grid_matches_oracle(ctx, raw):
    """pcl::VoxelGrid restated (LM:1044-1058 uses leaves 0.2 and 0.4): bit-exact, including a cloud
    larger than the LDS sort window and one with NaNs."""
    full = [O.sr_extract(s)["full"] for s in raw[:2]]
    clouds = [full[0][:3000], full[1], full[0][::7].copy(), np.zeros((0, 4), dtype=np.float32)]
    clouds[2][::50, 1] = np.nan
    for leaf in (0.2, 0.4):
        got = ctx.voxel_grid(clouds, leaf)
        for c, g in zip(clouds, got):
            ref, rc = O.voxel_grid(c, leaf)
            assert rc == 0
            assert g.shape == ref.shape and np.array_equal(g, ref)
    assert len(got[1]) < len(clouds[1]) // 3
    # the radix-sort path (key sets above the LDS window) with non-finite points in it, and a cloud of two sweeps
    holes = full[1].copy()
    holes[::97, 0] = np.nan
    holes[5::211, 2] = np.inf
    two = np.concatenate([full[0], full[1] + np.float32(0.05)])
    for leaf in (0.2, 0.4):
        got = ctx.voxel_grid([holes, two], leaf)
        for c, g in zip([holes, two], got):
            ref, rc = O.voxel_grid(c, leaf)
            assert rc == 0
            assert g.shape == ref.shape and np.array_equal(g, ref)
    # a single point, and leaves larger than the cloud (a voxel boundary still runs through the origin)
    rng = np.random.default_rng(2)
    odd = [rng.normal(0, 1, (1, 4)).astype(np.float32), rng.normal(0, 5, (5000, 4)).astype(np.float32)]
    for leaf in (50.0, 1e6):
        got = ctx.voxel_grid(odd, leaf)
        for c, g in zip(odd, got):
            ref, rc = O.voxel_grid(c, leaf)
            assert rc == 0 and g.shape == ref.shape and np.array_equal(g, ref)
