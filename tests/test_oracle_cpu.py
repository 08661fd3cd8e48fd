"""CPU tests (no GPU): the oracle against the fixtures and independent mathematics.

The reference ships no tests or golden vectors.  What pins the
oracle: (1) the known-answer values recorded in SURVEY.md section 8(c) from a probe of
the reference's own gps_process.cc on data/original_gps_data.txt, (2) numpy's SVD,
(3) an independent Krueger-series transverse Mercator, (4) brute force for the kd-tree,
(5) algebraic identities of the track fit.
"""
import math
import os

import numpy as np
import pytest

import _oracle as O
from gpscalibration_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------- reference KAT
def test_survey_kat_gps_to_enu_and_back(gps_log_bytes):
    """1000 stamps -> ENU (3450164.856218, 400633250.787481), back to
    121.398330784, 31.177944836, KML line 121.398330784171,31.177944836485,10."""
    st = 1494650700.0 + np.arange(1000)
    slam = np.zeros((1000, 4))
    slam[:, 2] = 10
    slam[:, 3] = st
    lat, lon, t = O.parse_gprmc(gps_log_bytes, st[0], st[-1])
    assert len(lat) == 1002  # fixes within [(long)(t0-1), (long)(t1+1)]
    enu = O.gps_to_enu(lat, lon, t, slam)
    assert len(enu) == 1000
    import json
    with open(os.path.join(GOLDEN, "survey_known_answers.json")) as f:
        kat = json.load(f)  # what the reference's own gps_process.cc printed (golden/README.md)
    assert "%.6f" % enu[0, 0] == "%.6f" % kat["enu_first"][0] == "3450164.856218"
    assert "%.6f" % enu[0, 1] == "%.6f" % kat["enu_first"][1] == "400633250.787481"
    ll, alt = O.local_to_wgs(np.c_[enu, np.ones(len(enu))])
    assert "%.9f %.9f" % (ll[0, 0], ll[0, 1]) == "%.9f %.9f" % tuple(kat["wgs84_back_first"])
    text = O.kml(ll, alt, 0)
    assert kat["kml_line_first"] + "\n" in text
    w = O.weights_speed(np.c_[enu[:, :2], np.zeros(len(enu)), st])
    assert w[-1] == kat["last_speed_weight"]


def test_parse_shipped_log_shape(gps_log_bytes):
    lat, lon, t = O.parse_gprmc(gps_log_bytes, 1494650697.0, 1494653187.0)
    assert len(lat) == 2490
    assert np.all(np.diff(t) > 0)
    assert abs(lat[0] - (31 + 10.67508 / 60)) < 1e-12
    assert abs(lon[0] - (121 + 23.90009 / 60)) < 1e-12
    assert not np.any((lat == 90) & (lon == 180))  # all status A


def test_parse_status_v_and_hemispheres():
    txt = ("100.5,$GPRMC,000001.00,A,3110.00000,S,12130.00000,W,0.1,,130517,,,A*00\n\n"
           "101.5,$GPRMC,000002.00,V,,,,,,,130517,,,N*00\n\n"
           "102.5,$GPRMC,000003.00,A,3111.00000,N,12131.00000,E,0.1,,130517,,,A*00\n")
    lat, lon, t = O.parse_gprmc(txt, 100.0, 103.0)
    assert list(t) == [100.5, 101.5, 102.5]
    assert lat[0] == -(31 + 10.0 / 60) and lon[0] == -(121 + 30.0 / 60)
    assert (lat[1], lon[1]) == (90.0, 180.0)  # sentinel, gps_process.cc:169,176-179
    la, lo, rc = O.gap_fill(lat, lon, t)
    assert rc == 0 and abs(la[1] - 0.5 * (lat[0] + lat[2])) < 1e-12


def test_gap_fill_cases():
    t = np.arange(8.0)
    lat = np.array([1.0, 2.0, 90, 90, 5.0, 6.0, 90, 90])
    lon = np.array([10.0, 20.0, 180, 180, 50.0, 60.0, 180, 180])
    la, lo, rc = O.gap_fill(lat, lon, t)
    assert rc == 0
    np.testing.assert_allclose(la, [1, 2, 3, 4, 5, 6, 7, 8], atol=1e-12)  # middle + trailing
    np.testing.assert_allclose(lo, [10, 20, 30, 40, 50, 60, 70, 80], atol=1e-11)
    lat = np.array([90, 90, 3.0, 4.0, 5.0])
    lon = np.array([180, 180, 30.0, 40.0, 50.0])
    la, lo, rc = O.gap_fill(lat, lon, np.arange(5.0))
    np.testing.assert_allclose(la, [1, 2, 3, 4, 5], atol=1e-12)  # leading gap: back-extrapolated
    la, lo, rc = O.gap_fill(np.array([1.0, 90, 90]), np.array([1.0, 180, 180]), np.arange(3.0))
    assert rc == 1  # one begin point only: cannot interpolate (gps_process.cc:444-447)


# ---------------------------------------------------------------- projection
def _krueger_tm(lat_deg, lon_deg, lon0_deg, k0, a=6378137.0, b=6356752.314):
    """Independent transverse Mercator (Krueger n-series, 6th order)."""
    f = (a - b) / a
    n = f / (2 - f)
    A = a / (1 + n) * (1 + n ** 2 / 4 + n ** 4 / 64 + n ** 6 / 256)
    al = [n / 2 - 2 * n ** 2 / 3 + 5 * n ** 3 / 16 + 41 * n ** 4 / 180,
          13 * n ** 2 / 48 - 3 * n ** 3 / 5 + 557 * n ** 4 / 1440,
          61 * n ** 3 / 240 - 103 * n ** 4 / 140,
          49561 * n ** 4 / 161280]
    phi, lam = math.radians(lat_deg), math.radians(lon_deg - lon0_deg)
    e = math.sqrt(f * (2 - f))
    t = math.sinh(math.atanh(math.sin(phi)) - e * math.atanh(e * math.sin(phi)))
    xi = math.atan2(t, math.cos(lam))
    eta = math.atanh(math.sin(lam) / math.sqrt(1 + t * t))
    N = xi + sum(al[j] * math.sin(2 * (j + 1) * xi) * math.cosh(2 * (j + 1) * eta) for j in range(4))
    E = eta + sum(al[j] * math.cos(2 * (j + 1) * xi) * math.sinh(2 * (j + 1) * eta) for j in range(4))
    return k0 * A * N, k0 * A * E


def test_utm_forward_close_to_krueger():
    # The reference's series (with its truncated PI and the misplaced A^6 term) must agree
    # with an exact transverse Mercator to a few millimetres inside a 3-degree band.
    lat = np.array([31.1779, 31.5, 30.2, -12.3, 45.0])
    lon = np.array([121.3983, 120.2, 119.1, 120.9, 121.4])
    xy = O.wgs_to_local(lat, lon, 0, 3)
    for i in range(len(lat)):
        n, e = _krueger_tm(lat[i], lon[i], 120.0, 0.9996)
        assert abs(xy[i, 0] - n) < 0.02, (i, xy[i, 0] - n)
        assert abs(xy[i, 1] - (e + 500000 + 40 * 1e7)) < 0.02


def test_gauss_forward_on_the_shipped_log_close_to_krueger(gps_log_bytes):
    # "Gaussion" (gps_process.cc:953-1007) on fixes of the log the reference ships, against the independent
    # Krueger series with k0 = 1: northing and (easting + 500 km + band * 10^7) within a centimetre.
    lat, lon, _ = O.parse_gprmc(gps_log_bytes, 1494650697.0, 1494653187.0)
    pick = np.arange(0, len(lat), 311)
    xy = O.wgs_to_local(lat[pick], lon[pick], 1, 3)
    for k, i in enumerate(pick):
        n, e = _krueger_tm(lat[i], lon[i], 120.0, 1.0, b=6356752.3142)
        assert abs(xy[k, 0] - n) < 0.01, (i, xy[k, 0] - n)
        assert abs(xy[k, 1] - (e + 500000 + 40 * 1e7)) < 0.01, (i, xy[k, 1] - e)


def test_irls_weights_known_answer():
    # weight_calculation.cc:30-78 by hand on a 4-point track: speed weights 1 (first point), |p2 - p1| / 2.2 =
    # 1.1 / 2.2 = 0.5, min(4.4 / 2.2, 1) = 1, last point = its distance from the origin / 2.2 capped at 1
    # (SURVEY 8c); residuals |E - F| = 0.5, 0.005 (clamped to 0.01), 2, 0.25 -> factors 2, 100, 0.5, 4.
    slam = np.array([[0, 0, 10, 0], [0.55, 0, 10, 1], [0.55, 1.1, 10, 2], [0.55, 5.5, 10, 3]], dtype=np.float64)
    fit = np.array([[5, 7, 10, 0], [6, 7, 10, 1], [6, 8, 10, 2], [6, 12, 10, 3]], dtype=np.float64)
    enu = fit.copy()
    enu[:, 0] += [0.5, 0.005, 2.0, 0.25]
    assert O.weights_speed(slam).tolist() == [1.0, 0.5, 1.0, 1.0]
    assert O.weights_irls(slam, enu, fit).tolist() == KNOWN_IRLS


KNOWN_IRLS = [2.0, 50.0, 0.5, 4.0]


def test_projection_round_trip_all_methods():
    rng = np.random.default_rng(1)
    lat = 31.0 + rng.uniform(-0.5, 0.5, 200)
    lon = 121.4 + rng.uniform(-0.3, 0.3, 200)
    for method in (0, 1):
        for band in (3, 6):
            xy = O.wgs_to_local(lat, lon, method, band)
            enu = np.c_[xy, np.full(200, 10.0), np.zeros(200), np.ones(200)]
            ll, alt = O.local_to_wgs(enu, method, band)
            assert np.abs(ll[:, 0] - lon).max() < 2e-8, (method, band)
            assert np.abs(ll[:, 1] - lat).max() < 2e-8
            assert np.all(alt == 10.0)


def test_interpolate_semantics():
    xy = np.array([[0.0, 0.0], [10.0, 100.0], [20.0, 200.0]])
    gt = np.array([10.0, 11.0, 12.0])
    st = np.array([9.5, 10.0, 10.25, 11.0, 11.5, 12.0, 12.5])
    out = O.interpolate(xy, gt, st)
    assert len(out) == 6  # 12.5 dropped (gps_process.cc:99), 9.5 extrapolated
    np.testing.assert_allclose(out[:, 0], [-5, 0, 2.5, 10, 15, 20], atol=1e-12)


# --------------------------------------------------------------------- SVD
def test_svd3_against_numpy():
    rng = np.random.default_rng(0)
    for i in range(500):
        A = rng.normal(size=(3, 3)) * 10 ** rng.uniform(-3, 3)
        if i % 5 == 0:
            A[2, :] = 0
            A[:, 2] = 0
        U, S, V = O.svd3(A)
        assert np.abs(U @ np.diag(S) @ V.T - A).max() <= 1e-13 * max(np.abs(A).max(), 1e-300)
        np.testing.assert_allclose(S, np.linalg.svd(A, compute_uv=False), rtol=1e-12, atol=1e-13 * S[0])
        assert np.abs(U.T @ U - np.eye(3)).max() < 1e-13 and np.abs(V.T @ V - np.eye(3)).max() < 1e-13
        if i % 5 == 0:  # zero third row/col keeps e_z in both factors (track path relies on it)
            assert abs(U[2, 2]) == 1.0 and abs(V[2, 2]) == 1.0


def test_kabsch_reflection_fix_matches_numpy():
    rng = np.random.default_rng(2)
    for _ in range(200):
        H = rng.normal(size=(3, 3))
        U, S, Vt = np.linalg.svd(H)
        R = Vt.T @ U.T
        if np.linalg.det(R) < 0:
            V = Vt.T.copy()
            V[:, 2] *= -1
            R = V @ U.T
        assert np.abs(O.kabsch(H) - R).max() < 1e-12


# ------------------------------------------------------------------- track
def _segment(n=400, seed=0, theta=0.7, noise=2.0):
    rng = np.random.default_rng(seed)
    xy, _ = synth.smooth_path(n, 0.1, seed)
    t = 1000.0 + 0.1 * np.arange(n)
    c, s = math.cos(theta), math.sin(theta)
    enu = np.c_[3450000 + xy[:, 0] + rng.normal(0, noise, n), 400633000 + xy[:, 1] + rng.normal(0, noise, n),
                np.full(n, 10.0), t]
    loc = xy - xy[0]
    slam = np.c_[c * loc[:, 0] - s * loc[:, 1], s * loc[:, 0] + c * loc[:, 1], np.full(n, 10.0), t]
    return slam, enu


def test_speed_weights_rule():
    slam, _ = _segment(50)
    w = O.weights_speed(slam)
    assert w[0] == 1.0
    d = np.hypot(*(slam[2:, :2] - slam[1:-1, :2]).T)
    np.testing.assert_allclose(w[1:-1], np.minimum(d / 2.2, 1.0), rtol=1e-15)
    # last slot: the out-of-bounds read is defined as (0,0)
    assert w[-1] == min(np.hypot(slam[-1, 0], slam[-1, 1]) / 2.2, 1.0)


def test_track_fit_recovers_rigid_motion():
    slam, enu = _segment(400, noise=0.0)
    w = np.ones(len(slam))
    T, rot, cal, iters = O.track_fit(slam, enu, w)
    assert iters in (1, 2)
    # noise-free: rotated SLAM + E0 reproduces the ENU track, and so does the calibrated track
    assert np.abs(rot[:, 0] + enu[0, 0] - enu[:, 0]).max() < 1e-6
    assert np.abs(cal[:, :2] - enu[:, :2]).max() < 1e-6
    assert np.all(cal[:, 2:] == enu[:, 2:])
    assert abs(np.linalg.det(T[:2, :2]) - 1) < 1e-12 and T[2, 2] == 1.0


def test_track_fit_quadratic_equals_closed_form():
    slam, enu = _segment(300, seed=3)
    w = O.weights_speed(slam)
    _, r1, c1, _ = O.track_fit(slam, enu, w, quadratic=True)
    _, r2, c2, _ = O.track_fit(slam, enu, w, quadratic=False)
    assert np.array_equal(r1, r2)
    assert np.abs(c1 - c2).max() < 1e-7  # y ~ 4e8: one ulp is 6e-8


def test_track_fit_reflection_case():
    # mirrored SLAM track: det H2 < 0 -> the reference keeps V2 U2^T (a reflection) in xy and
    # only flips R(2,2)
    slam, enu = _segment(200, seed=5, noise=0.0)
    slam[:, 1] *= -1
    T, rot, cal, _ = O.track_fit(slam, enu, np.ones(len(slam)))
    assert abs(np.linalg.det(T[:2, :2]) + 1) < 1e-12
    assert T[2, 2] == -1.0 and T[2, 3] == 2.0
    assert np.all(rot[:, 2] == 1.0)
    assert np.abs(cal[:, :2] - enu[:, :2]).max() < 1e-6


def test_long_segment_downweights_outliers():
    slam, enu = _segment(500, seed=9, noise=0.5)
    enu[100:110, :2] += 30.0  # multipath burst
    w, fit = O.long_segment(slam, enu, 5)
    assert np.median(w[100:110]) < 0.2 * np.median(w[200:400])
    assert w.shape == (500,) and fit.shape == (500, 4)


def test_match_and_merge_short():
    n = 30
    gps = np.c_[np.arange(n) * 1.0, np.zeros(n), np.full(n, 10.0), 100.0 + np.arange(n), np.linspace(1, 2, n)]
    slam = np.c_[np.zeros(10), np.zeros(10), np.full(10, 10.0), 105.0 + np.arange(10)]
    so, go, wo = O.match_gps(gps, slam)
    assert len(so) == 10 and np.all(go[:, 3] == slam[:, 3]) and np.all(wo == gps[5:15, 4])
    a = np.c_[np.arange(10.0), np.zeros(10), np.full(10, 10.0), 100.0 + np.arange(10)]
    acc = O.merge_short(None, a, np.ones(10))
    b = np.c_[np.arange(6, 16.0) + 1.0, np.zeros(10), np.full(10, 10.0), 106.0 + np.arange(10)]
    acc = O.merge_short(acc, b, np.full(10, 3.0))
    assert len(acc) == 16
    assert np.all(acc[:6, 0] == np.arange(6.0))  # untouched head
    assert np.all(acc[10:, 0] == np.arange(10, 16.0) + 1.0)  # appended tail
    # overlap of 4: opNo=4, smWindow=2 -> new-segment share coe2 = 1/4, 2/4, 2/4, 3/4
    # (short_distance_track_process.cpp:110-124: ramp in, plateau, ramp out of the OLD track)
    np.testing.assert_allclose(acc[6:10, 0] - np.arange(6, 10.0), [0.25, 0.5, 0.5, 0.75])
    np.testing.assert_allclose(acc[6:10, 4], 1 + 2 * np.array([0.25, 0.5, 0.5, 0.75]))


def test_colour_segments_and_kml_quirks():
    n = 40
    enu = np.c_[np.arange(n) * 10.0, np.zeros(n), np.full(n, 10.0), np.arange(n) * 1.0, np.full(n, 0.5)]
    end, rgb = O.colour_segments(enu)
    assert list(end[:2]) == [6, 12]  # first index where the running distance exceeds 50 m
    assert end[-1] == n - 1
    ll = np.c_[121 + np.arange(n) * 1e-4, 31 + np.zeros(n)]
    alt = np.full(n, 10.0)
    txt = O.kml(ll, alt, 1, end, rgb)
    assert txt.count("<Placemark>") == len(end)
    coords = [l for l in txt.splitlines() if l.count(",") == 2 and l[0].isdigit()]
    assert len(coords) == n - 1  # calibrated KML never writes the last point (gps_process.cc:832)
    txt0 = O.kml(ll, alt, 0)
    assert len([l for l in txt0.splitlines() if l.count(",") == 2 and l[0].isdigit()]) == n


def test_height_compensation_keeps_step_length():
    rng = np.random.default_rng(4)
    p = np.cumsum(rng.normal(0, 1, size=(50, 3)), axis=0)
    loam = np.c_[p, 10.0 + np.arange(50)]
    out = O.height_compensate(loam)
    d3 = np.linalg.norm(np.diff(p, axis=0), axis=1)
    d2 = np.hypot(*np.diff(out[:, :2], axis=0).T)
    np.testing.assert_allclose(d2, d3, rtol=1e-9)
    assert np.all(out[:, 2] == 10.0)


# ----------------------------------------------------------------- kNN/ICP
@pytest.mark.parametrize("k", [1, 5])
def test_kdtree_equals_brute_force(k):
    rng = np.random.default_rng(0)
    tgt = (rng.normal(size=(4000, 3)) * 10).astype(np.float32)
    tgt[100:200] = tgt[0:100]  # exact duplicates: ties go to the lower index
    q = (rng.normal(size=(1500, 3)) * 12).astype(np.float32)
    q[:50] = tgt[:50]
    i1, d1 = O.knn_brute(tgt, q, k)
    i2, d2 = O.KdTree(tgt).search(q, k)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    assert np.all(i1[:50, 0] == np.arange(50)) and np.all(d1[:50, 0] == 0)


def test_knn_edge_cases():
    tgt = np.array([[0, 0, 0], [1, 0, 0]], dtype=np.float32)
    i, d = O.knn_brute(tgt, np.array([[0.4, 0, 0]], dtype=np.float32), 5)
    assert list(i[0]) == [0, 1, -1, -1, -1] and np.isinf(d[0, 2])
    i, d = O.KdTree(tgt).search(np.zeros((0, 3), dtype=np.float32), 1)
    assert i.shape == (0, 1)


def test_icp_recovers_transform():
    tgt, src, T_true = synth.scan_pair(8192, 0)
    kd = O.KdTree(tgt)
    T, hist = kd.icp_run(src, 30)
    assert hist[-1] < hist[0]
    assert np.abs(T[:3, :3] - T_true[:3, :3]).max() < 2e-3
    assert np.abs(T[:3, 3] - T_true[:3, 3]).max() < 0.05


# ------------------------------------------------------------------ LOAM loops
def test_lo_oracle_recovers_sweep_motion():
    """laserOdometry.cpp:585-1029 restated: the loop must pull the sweep-to-sweep transform towards
    the motion the synthetic sweep was generated with (forward motion -> negative tz).  The steps
    are damped (b = -0.05 d, LO:970), so 25 iterations do not finish the job: a second call warm-started
    at the first answer keeps moving the same way, as the next sweep's initial guess does in LOAM."""
    from gpscalibration_amd import synth
    A = synth.loam_sweep((0, 0, 0, 0, 0, 0), seed=1)
    B = synth.loam_sweep((0.004, 0.015, -0.003, 0.05, 0.01, 0.45), seed=10)
    tr, it, ns = O.lo_match(B["sharp"], B["flat"], A["less_sharp"], A["less_flat"])
    assert 2 <= it <= 25 and ns > 100
    assert tr[5] < -0.1
    tr2, it2, _ = O.lo_match(B["sharp"], B["flat"], A["less_sharp"], A["less_flat"], tr)
    assert it2 <= 25 and -0.5 < tr2[5] < tr[5]


def test_lm_oracle_converges_to_the_map_pose():
    """laserMapping.cpp:748-1018 restated: features of a sweep taken at the map's own pose, started
    from a perturbed transformTobeMapped, come back to (near) zero."""
    from gpscalibration_amd import synth
    M = synth.loam_sweep((0, 0, 0, 0, 0, 0), seed=3, n_az=1800)
    B = synth.loam_sweep((0, 0, 0, 0, 0, 0), seed=2)
    tr0 = np.array([0.003, -0.01, 0.002, 0.08, -0.02, 0.12], dtype=np.float32)
    tr, it, ns = O.lm_match(B["less_sharp"], B["flat"], M["less_sharp"], M["less_flat"], tr0)
    assert 2 <= it <= 10 and ns >= 50
    assert np.abs(tr[:3]).max() < 2e-3 and np.abs(tr[3:]).max() < 2e-2
    # too small a map: untouched (LM:748)
    tr, it, _ = O.lm_match(B["less_sharp"], B["flat"], M["less_sharp"][:10], M["less_flat"], tr0)
    assert it == 0 and np.array_equal(tr, tr0)


# ------------------------------------------------------- scanRegistration / VoxelGrid
def test_sr_oracle_invariants():
    """scanRegistration.cpp:238-674 restated: structural facts the reference's code guarantees."""
    from gpscalibration_amd import synth
    W = synth.lidar_world(0)
    P = synth.raw_sweep(W, pos=(10, 0.5), yaw=0.05, vel=(8, 0), seed=1, nan_every=997)
    F = O.sr_extract(P)
    full = F["full"]
    ring = full[:, 3].astype(int)
    assert len(full) <= np.isfinite(P).all(axis=1).sum()
    assert (np.diff(ring) >= 0).all() and ring.min() == 0 and ring.max() == 15  # SR:444-447
    rel = (full[:, 3] - ring) * 10
    assert rel.min() > -1e-3 and rel.max() < 1.0 + 1e-3  # SR:361-362
    # axis swap SR:295-297: LOAM y is up, so the ground (sensor z = -1.8) sits at y = -1.8
    assert abs(np.median(full[ring == 0, 1]) + 1.8) < 0.05
    # per (ring, sector) caps SR:583-592, 629
    assert len(F["sharp"]) <= 16 * 6 * 16 and len(F["less_sharp"]) <= 16 * 6 * 20 and len(F["flat"]) <= 16 * 6 * 32
    # every sharp point is also less sharp, in order (SR:585-587)
    ls = {tuple(p) for p in F["less_sharp"]}
    assert all(tuple(p) in ls for p in F["sharp"])
    # features are points of the full cloud
    fs = {tuple(p) for p in full}
    assert all(tuple(p) in fs for p in F["flat"]) and all(tuple(p) in fs for p in F["sharp"])
    # poles and building edges give corners; ground and walls give flats
    assert len(F["sharp"]) > 100 and len(F["flat"]) > 2000


def test_voxel_grid_oracle_is_a_centroid_filter():
    """pcl::VoxelGrid restated: one output per occupied 0.4 m cell, equal to the mean of its points,
    ordered by cell id (x fastest)."""
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform(-3, 3, (4000, 3)), rng.uniform(0, 16, (4000, 1))], axis=1).astype(np.float32)
    out, rc = O.voxel_grid(pts, 0.4)
    assert rc == 0
    inv = np.float32(1.0) / np.float32(0.4)
    ijk = np.floor(pts[:, :3] * inv).astype(np.int64)
    ijk -= np.floor(pts[:, :3].min(axis=0) * inv).astype(np.int64)
    div = ijk.max(axis=0) + 1
    cell = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    ids = np.unique(cell)
    assert len(out) == len(ids)
    means = np.stack([pts[cell == c].astype(np.float64).mean(axis=0) for c in ids])
    assert np.abs(out - means).max() < 1e-4
    # far-apart points blow the cell count past INT_MAX: PCL gives the input back
    far = np.array([[0, 0, 0, 1], [1e5, 1e5, 1e5, 2]], dtype=np.float32)
    out, rc = O.voxel_grid(far, 0.2)
    assert rc == 1 and np.array_equal(out, far)


def test_voxel_grid_of_a_growing_map_is_a_merge():
    """pcl::VoxelGrid re-applied to (its own output + new points), as laserMapping does per map cube (LM:1044-1078):
    the old part is one centroid per voxel in voxel order, so the filter equals a stable merge of it with the sorted
    new points -- the property an incremental device filter may rely on (checked against the full filter here)."""
    from gpscalibration_amd import synth
    W = synth.lidar_world(0, length=200.0)
    sw, _, truth = synth.drive(W, 7, seed=40, n_az=450, start=(0.0, 0.0))
    leaf = np.float32(0.4)
    inv = np.float32(1.0) / leaf
    old = np.zeros((0, 4), dtype=np.float32)
    for t, s in enumerate(sw):
        new = O.sr_extract(s)["less_flat"].copy()[::2]
        new[:, :3] += np.float32(truth[t, :3])
        cloud = np.concatenate([old, new]).astype(np.float32)
        ref, rc = O.voxel_grid(cloud, float(leaf))
        assert rc == 0
        if len(old):
            lo, hi = cloud[:, :3].min(0), cloud[:, :3].max(0)
            minb = np.floor(lo * inv).astype(np.int64)
            div = np.floor(hi * inv).astype(np.int64) - minb + 1

            def keys(p):
                ijk = (np.floor(p[:, :3] * inv) - minb.astype(np.float32)).astype(np.int64)
                return ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]

            ko, kn = keys(old), keys(new)
            assert (np.diff(ko) > 0).all()  # sorted, one point per voxel
            order = np.argsort(kn, kind="stable")
            allk = np.concatenate([ko, kn[order]])
            allp = np.concatenate([old, new[order]])
            side = np.concatenate([np.zeros(len(ko), np.int8), np.ones(len(kn), np.int8)])
            idx = np.lexsort((np.arange(len(allk)), side, allk))  # old before new inside a voxel, new in input order
            allk, allp = allk[idx], allp[idx]
            starts = np.flatnonzero(np.r_[True, np.diff(allk) != 0])
            ends = np.r_[starts[1:], len(allk)]
            got = np.empty((len(starts), 4), np.float32)
            for r, (a, b) in enumerate(zip(starts, ends)):
                acc = np.zeros(4, np.float32)
                for e in range(a, b):
                    acc = (acc + allp[e]).astype(np.float32)
                got[r] = acc / np.float32(b - a)
            assert got.shape == ref.shape and np.array_equal(got, ref), t
        old = ref
    assert len(old) > 2000


def test_certificates_with_a_second_neighbour_list_and_packed_radii():
    """The step kernel's certificates (a query closer than half of D1 / D5 to last iteration's
    neighbour q0 has its nearest neighbour at q0 / among q0 and its 4 nearest) extended by a second list of four
    (half of D9), with the three squared radii packed into the 32 bits the warm stream has today: 16 truncated bits
    of r_a^2 and two 8-bit log2 ratios rounded down.  The decoded radii never exceed the exact ones, and every tier
    returns the exact neighbour (k-d tree of scipy as the judge)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(5)
    m = 6000
    tgt = np.r_[np.c_[rng.uniform(-10, 10, (m // 2, 2)), rng.normal(0, 0.01, m // 2)],
                np.c_[rng.uniform(-10, 10, m // 2), 5.0 + rng.normal(0, 0.01, m // 2), rng.uniform(0, 4, m // 2)]]
    tgt = tgt.astype(np.float32)
    tree = cKDTree(tgt.astype(np.float64))
    dk, ik = tree.query(tgt.astype(np.float64), k=10)
    r2 = [(np.float32(0.25 * 0.99) * dk[:, k].astype(np.float32) ** 2).astype(np.float32) for k in (1, 5, 9)]
    ra_bits = r2[0].view(np.uint32) & np.uint32(0xffff0000)

    def enc(x):
        ratio = np.maximum(x.astype(np.float64) / np.maximum(r2[0].astype(np.float64), 1e-300), 1.0)
        return np.clip(np.floor(np.log2(ratio) * 32.0 - 0.01), 0, 255).astype(np.uint32)

    word = ra_bits | (enc(r2[1]) << np.uint32(8)) | enc(r2[2])
    ra = (word & np.uint32(0xffff0000)).view(np.float32)
    rb = (ra * np.exp2(((word >> np.uint32(8)) & np.uint32(0xff)).astype(np.float32) / np.float32(32))).astype(np.float32)
    rc = (ra * np.exp2((word & np.uint32(0xff)).astype(np.float32) / np.float32(32))).astype(np.float32)
    assert (ra <= r2[0]).all() and (rb <= r2[1]).all() and (rc <= r2[2]).all()
    assert np.median(1 - np.sqrt(rc / r2[2])) < 0.02  # what the packing gives away

    nq = 20000
    q = (tgt[rng.integers(0, m, nq)] + rng.normal(0, 0.08, (nq, 3))).astype(np.float32)
    _, q0 = tree.query((q + rng.normal(0, 0.05, q.shape)).astype(np.float64))
    dtrue, _ = tree.query(q.astype(np.float64))
    d0 = ((q - tgt[q0]) ** 2).sum(1).astype(np.float32)
    t1 = d0 < ra[q0]
    t2 = ~t1 & (d0 < rb[q0])
    t3 = ~t1 & ~t2 & (d0 < rc[q0])
    assert t1.mean() > 0.1 and t2.mean() > 0.1 and t3.mean() > 0.05  # every tier is exercised

    def nearest_of(cands):
        dd = ((tgt[cands].astype(np.float64) - q[:, None, :].astype(np.float64)) ** 2).sum(2)
        return np.sqrt(dd.min(1))

    for msk, cands in ((t1, q0[:, None]), (t2, ik[q0][:, :5]), (t3, ik[q0][:, :9])):
        np.testing.assert_allclose(nearest_of(cands)[msk], dtrue[msk], rtol=1e-9, atol=1e-12)


def test_input_data_oracle_cuts_tracks_by_distance():
    """input_data.cpp:78-124, 266-444 restated around the node chain: tracks are cut when the travelled
    distance exceeds the segment length, the next one restarts after the last sample within
    length - overlap, and the first message after every restart publishes no odometry (LO:519-562)."""
    from gpscalibration_amd import synth
    W = synth.lidar_world(0, length=600.0)
    sw, st, _ = synth.drive(W, 60, seed=1, n_az=450)
    longs = O.input_data_pass(sw, st, 25.0, 0.0)
    shorts = O.input_data_pass(sw, st, 12.0, 4.0)
    for tracks, length, ov in ((longs, 25.0, 0.0), (shorts, 12.0, 4.0)):
        assert tracks[0]["first"] == 1 and tracks[-1]["last"] == 60
        for a, b in zip(tracks, tracks[1:]):
            assert a["first"] < b["first"] <= a["last"]  # restart inside the previous track
        for t in tracks:
            n_msgs = t["last"] - t["first"] + 1
            assert len(t["track"]) == n_msgs - 1  # the restarting message has no odometry
            d = np.hypot(np.diff(t["track"][:, 0]), np.diff(t["track"][:, 1])).sum()
            assert d < length + 3.0  # cut right after the crossing sample
            assert np.all(t["track"][:, 2] == 10.0) and np.all(np.diff(t["track"][:, 3]) > 0)
    assert len(shorts) > len(longs) >= 2
    # short tracks overlap by roughly the overlap distance
    assert any(b["first"] < a["last"] for a, b in zip(shorts, shorts[1:]))


# ------------------------------------------------ alternate sentences and map datums
def _gga(t, lat, lon, ok=True):
    la, lo = abs(lat), abs(lon)
    f = lambda v, w: ("%0" + str(w) + ".5f") % (int(v) * 100 + (v - int(v)) * 60)
    if not ok:
        return "%.8f,$GPGGA,044500.00,,,,,0,00,99.99,,,,,,*48" % t
    return "%.8f,$GPGGA,044500.00,%s,%s,%s,%s,1,08,1.0,10.0,M,8.0,M,,*5A" % (
        t, f(la, 10), "N" if lat >= 0 else "S", f(lo, 11), "E" if lon >= 0 else "W")


def _gll(t, lat, lon, status="A"):
    la, lo = abs(lat), abs(lon)
    f = lambda v, w: ("%0" + str(w) + ".5f") % (int(v) * 100 + (v - int(v)) * 60)
    return "%.8f,$GPGLL,%s,%s,%s,%s,044500.00,%s,A*6D" % (t, f(la, 10), "N" if lat >= 0 else "S", f(lo, 11),
                                                          "E" if lon >= 0 else "W", status)


def test_gpgga_and_gpgll_logs_parse_like_the_reference():
    """gps_process.cc:113-159 picks the parser from the first line; GPGGA (:231-299) drops fixes
    without coordinates, GPGLL (:300-372) keeps every line in the window, status V included."""
    t0 = 1494650700.0
    rows = [(t0 + k, 31.17 + 1e-4 * k, 121.39 + 2e-4 * k) for k in range(8)]
    gga = "\n".join(_gga(t, la, lo, ok=(k != 3)) for k, (t, la, lo) in enumerate(rows)) + "\n"
    lat, lon, t = O.parse_gps_log(gga, t0 + 1, t0 + 5)
    assert list(t) == [t0 + k for k in (0, 1, 2, 4, 5, 6)]  # window +-1 s (long casts), line 3 has no fix
    assert np.abs(lat - np.array([rows[k][1] for k in (0, 1, 2, 4, 5, 6)])).max() < 1e-7
    assert np.abs(lon - np.array([rows[k][2] for k in (0, 1, 2, 4, 5, 6)])).max() < 1e-7
    gll = "\n".join(_gll(t, -la, -lo, "V" if k == 2 else "A") for k, (t, la, lo) in enumerate(rows)) + "\n"
    lat, lon, t = O.parse_gps_log(gll, t0 + 1, t0 + 5)
    assert len(t) == 7 and lat[2] < 0 and lon[2] < 0  # southern / western hemisphere signs; V kept
    assert np.abs(lat + np.array([r[1] for r in rows[:7]])).max() < 1e-7
    # the shipped GPRMC log goes through the same entry point unchanged
    text = open(os.path.join(GOLDEN, "original_gps_data.txt")).read()
    a = O.parse_gps_log(text, 1494650900.0, 1494651000.0)
    b = O.parse_gprmc(text, 1494650900.0, 1494651000.0)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and len(a[2]) > 50
    assert len(O.parse_gps_log("1.0,$GPVTG,1,2\n", 0, 10)[2]) == 0  # unsupported sentence: nothing


def test_gcj02_bd09_transforms():
    """gps_process.cc:526-595, 1127-1207.  Known behaviour of the public 'Mars' offset: a few hundred
    metres inside China, identity outside; BD-09 adds ~0.006 deg; bd_decrypt undoes bd_encrypt to ~1e-5 deg."""
    ll = np.array([[121.398330784, 31.177944836], [116.3975, 39.9087], [2.2945, 48.8584], [139.69, 35.68]])
    gcj = O.mars(ll, "gps_to_gcj")
    d = gcj - ll
    assert 1e-3 < abs(d[0, 0]) < 1e-2 and 1e-3 < abs(d[0, 1]) < 1e-2  # Shanghai: ~0.0045, ~-0.002 deg
    assert np.array_equal(gcj[2], ll[2]) and np.array_equal(gcj[3], ll[3])  # Paris, Tokyo: outside the box
    bd = O.mars(gcj, "gcj_to_bd")
    assert np.all(np.abs(bd[:2] - gcj[:2] - [0.0065, 0.006]) < 2e-3)
    back = O.mars(bd, "bd_to_gcj")
    assert np.abs(back - gcj).max() < 2e-5
    # independent re-derivation of one value with the reference's truncated PI
    PI, a, ee = 3.141592653589, 6378245.0, (6378245.0 ** 2 - 6356863.0188 ** 2) / 6378245.0 ** 2
    x, y = ll[1, 0] - 105.0, ll[1, 1] - 35.0
    dlat = (-100.0 + 2.0 * x + 3.0 * y + 0.2 * y * y + 0.1 * x * y + 0.2 * math.sqrt(abs(x))
            + (20.0 * math.sin(6.0 * x * PI) + 20.0 * math.sin(2.0 * x * PI)) * 2.0 / 3.0
            + (20.0 * math.sin(y * PI) + 40.0 * math.sin(y / 3.0 * PI)) * 2.0 / 3.0
            + (160.0 * math.sin(y / 12.0 * PI) + 320 * math.sin(y * PI / 30.0)) * 2.0 / 3.0)
    rad = ll[1, 1] / 180.0 * PI
    magic = 1 - ee * math.sin(rad) ** 2
    dlat = (dlat * 180.0) / ((a * (1 - ee)) / (magic * math.sqrt(magic)) * PI)
    assert abs(gcj[1, 1] - (ll[1, 1] + dlat)) < 1e-12


def test_json_writer_layout():
    """createJSON, gps_process.cc:1210-1250: precision(15), trailing commas and all."""
    ll = np.array([[121.5, 31.25], [121.50000123456789, 31.2500009], [121.6, 31.3]])
    assert O.json_map(ll, 0) == ('[{"line":[[121.5,31.25],[121.500001234568,31.2500009],[121.6,31.3],],'
                                 '"color":"FF00FF"}]')
    got = O.json_map(ll, 1, [0, 2], [0xFF0000, 0x00FF7F])
    assert got == ('[{"line":[[121.5,31.25],],"color":"FF0000"},{"line":[[121.500001234568,31.2500009],'
                   '[121.6,31.3],],"color":"00FF7F"},]')
