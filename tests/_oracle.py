"""ctypes binding of oracle/liboracle.so -- the CPU restatement used as the checker.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by gpscalibration_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int32)
c_up = C.POINTER(C.c_uint32)


def _p(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        _LIB = C.CDLL(so)
        _LIB.orc_sqdist.restype = C.c_float
        _LIB.orc_kdtree_build.restype = C.c_void_p
        _LIB.orc_kml.restype = C.c_long
    return _LIB


# ------------------------------------------------------------------ weights
def weights_speed(slam):
    slam = f64(slam)
    n = len(slam)
    w = np.empty(n)
    lib().orc_weights_speed(_p(slam, c_dp), n, _p(w, c_dp))
    return w


def weights_irls(slam, enu, fit):
    slam, enu, fit = f64(slam), f64(enu), f64(fit)
    n = len(slam)
    w = np.empty(n)
    lib().orc_weights_irls(_p(slam, c_dp), _p(enu, c_dp), _p(fit, c_dp), n, _p(w, c_dp))
    return w


# ---------------------------------------------------------------------- svd
def svd3(A):
    A = f64(A).reshape(9)
    U, S, V = np.empty(9), np.empty(3), np.empty(9)
    lib().orc_svd3(_p(A, c_dp), _p(U, c_dp), _p(S, c_dp), _p(V, c_dp))
    return U.reshape(3, 3), S, V.reshape(3, 3)


def kabsch(H):
    H = f64(H).reshape(9)
    R = np.empty(9)
    lib().orc_kabsch_from_H(_p(H, c_dp), _p(R, c_dp))
    return R.reshape(3, 3)


# -------------------------------------------------------------------- track
def track_fit(slam, enu, w, quadratic=True):
    slam, enu, w = f64(slam), f64(enu), f64(w)
    n = len(slam)
    T = np.empty(16)
    rot = np.empty((n, 3))
    cal = np.empty((n, 4))
    it = C.c_int(0)
    rc = lib().orc_track_fit(_p(slam, c_dp), _p(enu, c_dp), _p(w, c_dp), n, _p(T, c_dp),
                             _p(rot, c_dp), _p(cal, c_dp), C.byref(it), int(quadratic))
    assert rc == 1
    return T.reshape(4, 4), rot, cal, it.value


def long_segment(slam, enu, irls_iters=5, quadratic=True):
    slam, enu = f64(slam), f64(enu)
    n = len(slam)
    w = np.empty(n)
    fit = np.empty((n, 4))
    rc = lib().orc_long_segment(_p(slam, c_dp), _p(enu, c_dp), n, irls_iters, _p(w, c_dp),
                                _p(fit, c_dp), int(quadratic))
    assert rc == 1
    return w, fit


def match_gps(gps_xyztw, slam_xyzt):
    gps, slam = f64(gps_xyztw), f64(slam_xyzt)
    n = len(slam)
    so, go, wo = np.empty((n, 4)), np.empty((n, 4)), np.empty(n)
    m = lib().orc_match_gps(_p(gps, c_dp), len(gps), _p(slam, c_dp), n, _p(so, c_dp),
                            _p(go, c_dp), _p(wo, c_dp))
    return so[:m].copy(), go[:m].copy(), wo[:m].copy()


def merge_short(acc, seg, segw):
    """acc: (na,5) or empty; returns the new accumulated track."""
    seg, segw = f64(seg), f64(segw)
    na = 0 if acc is None else len(acc)
    cap = na + len(seg) + 8
    buf = np.zeros((cap, 5))
    if na:
        buf[:na] = acc
    n = C.c_int(na)
    rc = lib().orc_merge_short(_p(buf, c_dp), C.byref(n), cap, _p(seg, c_dp), _p(segw, c_dp), len(seg))
    assert rc == 0
    return buf[:n.value].copy()


def height_compensate(loam_xyzt):
    p = f64(loam_xyzt)
    out = np.empty((len(p), 4))
    lib().orc_height_compensate(_p(p, c_dp), len(p), _p(out, c_dp))
    return out


# ---------------------------------------------------------------------- geo
def parse_gprmc(text, t0, t1):
    if isinstance(text, str):
        text = text.encode()
    cap = text.count(b"\n") + 2
    lat, lon, t = np.empty(cap), np.empty(cap), np.empty(cap)
    n = lib().orc_parse_gprmc(text, C.c_size_t(len(text)), C.c_double(t0), C.c_double(t1),
                              _p(lat, c_dp), _p(lon, c_dp), _p(t, c_dp), cap)
    assert n >= 0
    return lat[:n].copy(), lon[:n].copy(), t[:n].copy()


def gap_fill(lat, lon, t):
    lat, lon, t = f64(lat).copy(), f64(lon).copy(), f64(t)
    rc = lib().orc_gap_fill(_p(lat, c_dp), _p(lon, c_dp), _p(t, c_dp), len(t))
    return lat, lon, rc


def wgs_to_local(lat, lon, method=0, band_type=3):
    lat, lon = f64(lat), f64(lon)
    xy = np.empty((len(lat), 2))
    lib().orc_wgs_to_local(method, band_type, _p(lat, c_dp), _p(lon, c_dp), len(lat), _p(xy, c_dp))
    return xy


def local_to_wgs(enu_xyztw, method=0, band_type=3):
    e = f64(enu_xyztw)
    ll = np.empty((len(e), 2))
    alt = np.empty(len(e))
    lib().orc_local_to_wgs(method, band_type, _p(e, c_dp), len(e), _p(ll, c_dp), _p(alt, c_dp))
    return ll, alt


def interpolate(xy, gps_t, slam_t):
    xy, gps_t, slam_t = f64(xy), f64(gps_t), f64(slam_t)
    out = np.empty((len(slam_t), 2))
    m = lib().orc_interpolate(_p(xy, c_dp), _p(gps_t, c_dp), len(gps_t), _p(slam_t, c_dp),
                              len(slam_t), _p(out, c_dp))
    return out[:m].copy()


def gps_to_enu(lat, lon, gps_t, slam_xyzt, method=0, band_type=3):
    lat, lon, gps_t = f64(lat).copy(), f64(lon).copy(), f64(gps_t)
    slam = f64(slam_xyzt)
    enu = np.empty((len(slam), 4))
    m = lib().orc_gps_to_enu(method, band_type, _p(lat, c_dp), _p(lon, c_dp), _p(gps_t, c_dp),
                             len(gps_t), _p(slam, c_dp), len(slam), _p(enu, c_dp))
    assert m >= 0
    return enu[:m].copy()


def colour_segments(enu_xyztw):
    e = f64(enu_xyztw)
    cap = len(e) + 1
    end = np.empty(cap, dtype=np.int32)
    rgb = np.empty(cap, dtype=np.uint32)
    k = lib().orc_colour_segments(_p(e, c_dp), len(e), _p(end, c_ip), _p(rgb, c_up), cap)
    assert k >= 0
    return end[:k].copy(), rgb[:k].copy()


def kml(lonlat, alt, flag, seg_end=None, rgb=None):
    ll, alt = f64(lonlat), f64(alt)
    nseg = 0 if seg_end is None else len(seg_end)
    se = None if seg_end is None else np.ascontiguousarray(seg_end, dtype=np.int32)
    cc = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint32)
    need = lib().orc_kml(None, C.c_size_t(0), _p(ll, c_dp), _p(alt, c_dp), len(ll), flag,
                         _p(se, c_ip), _p(cc, c_up), nseg)
    buf = C.create_string_buffer(need + 1)
    lib().orc_kml(buf, C.c_size_t(need + 1), _p(ll, c_dp), _p(alt, c_dp), len(ll), flag,
                  _p(se, c_ip), _p(cc, c_up), nseg)
    return buf.value.decode()


# ------------------------------------------------------------------ knn/icp
def knn_brute(tgt, q, k=1):
    tgt, q = f32(tgt), f32(q)
    n = len(q)
    idx = np.empty((n, k), dtype=np.int32)
    sqd = np.empty((n, k), dtype=np.float32)
    lib().orc_knn_brute(_p(tgt, c_fp), len(tgt), _p(q, c_fp), n, k, _p(idx, c_ip), _p(sqd, c_fp))
    return idx, sqd


class KdTree:
    def __init__(self, tgt):
        self.tgt = f32(tgt)  # keep alive: the tree borrows the buffer
        self.h = C.c_void_p(lib().orc_kdtree_build(_p(self.tgt, c_fp), len(self.tgt)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kdtree_free(self.h)
            self.h = None

    def search(self, q, k=1):
        q = f32(q)
        n = len(q)
        idx = np.empty((n, k), dtype=np.int32)
        sqd = np.empty((n, k), dtype=np.float32)
        lib().orc_kdtree_search(self.h, _p(q, c_fp), n, k, _p(idx, c_ip), _p(sqd, c_fp))
        return idx, sqd

    def icp_iterate(self, src, T_in, w=None):
        src = f32(src)
        T_in = f64(T_in).reshape(16)
        n = len(src)
        T = np.empty(16)
        err = C.c_double(0)
        idx = np.empty(n, dtype=np.int32)
        sqd = np.empty(n, dtype=np.float32)
        wv = None if w is None else f64(w)
        rc = lib().orc_icp_iterate(self.h, _p(src, c_fp), n, _p(wv, c_dp), _p(T_in, c_dp),
                                   _p(T, c_dp), C.byref(err), _p(idx, c_ip), _p(sqd, c_fp))
        assert rc == 0
        return T.reshape(4, 4), err.value, idx, sqd

    def icp_run(self, src, iters, T0=None, w=None):
        src = f32(src)
        T0 = f64(np.eye(4) if T0 is None else T0).reshape(16)
        T = np.empty(16)
        hist = np.empty(iters)
        wv = None if w is None else f64(w)
        rc = lib().orc_icp_run(self.h, _p(src, c_fp), len(src), _p(wv, c_dp), iters,
                               _p(T0, c_dp), _p(T, c_dp), _p(hist, c_dp))
        assert rc == 0
        return T.reshape(4, 4), hist


def transform_f32(T, src):
    src = f32(src)
    T = f64(T).reshape(16)
    dst = np.empty_like(src)
    lib().orc_transform_f32(_p(T, c_dp), _p(src, c_fp), len(src), _p(dst, c_fp))
    return dst


# --------------------------------------------------------------------- LOAM
def lo_transform(tr, pts, to_end=False):
    tr = np.ascontiguousarray(tr, dtype=np.float32)
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty_like(pts)
    fn = lib().orc_lo_transform_to_end if to_end else lib().orc_lo_transform_to_start
    for i in range(len(pts)):
        fn(_p(tr, c_fp), C.c_void_p(pts.ctypes.data + 16 * i), C.c_void_p(out.ctypes.data + 16 * i))
    return out


def lo_match(sharp, flat, corner_last, surf_last, tr_in=None):
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (sharp, flat, corner_last, surf_last)]
    tr_in = np.zeros(6, dtype=np.float32) if tr_in is None else np.ascontiguousarray(tr_in, dtype=np.float32)
    tr = np.empty(6, dtype=np.float32)
    it, ns = C.c_int(0), C.c_int(0)
    lib().orc_lo_match(_p(a[0], c_fp), len(a[0]), _p(a[1], c_fp), len(a[1]), _p(a[2], c_fp), len(a[2]),
                       _p(a[3], c_fp), len(a[3]), _p(tr_in, c_fp), _p(tr, c_fp), C.byref(it), C.byref(ns))
    return tr, it.value, ns.value


def lo_accumulate(sum_in, tr):
    s = np.ascontiguousarray(sum_in, dtype=np.float32)
    t = np.ascontiguousarray(tr, dtype=np.float32)
    out = np.empty(6, dtype=np.float32)
    lib().orc_lo_accumulate(_p(s, c_fp), _p(t, c_fp), _p(out, c_fp))
    return out


def lm_match(corner_stack, surf_stack, corner_map, surf_map, tr_in=None):
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (corner_stack, surf_stack, corner_map, surf_map)]
    tr_in = np.zeros(6, dtype=np.float32) if tr_in is None else np.ascontiguousarray(tr_in, dtype=np.float32)
    tr = np.empty(6, dtype=np.float32)
    it, ns = C.c_int(0), C.c_int(0)
    lib().orc_lm_match(_p(a[0], c_fp), len(a[0]), _p(a[1], c_fp), len(a[1]), _p(a[2], c_fp), len(a[2]),
                       _p(a[3], c_fp), len(a[3]), _p(tr_in, c_fp), _p(tr, c_fp), C.byref(it), C.byref(ns))
    return tr, it.value, ns.value


def sr_extract(xyz):
    """scanRegistration.cpp:238-674 on one raw sweep [n,3] float32 -> dict of [k,4] clouds."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = len(xyz)
    names = ("full", "sharp", "less_sharp", "flat", "less_flat")
    bufs = {k: np.zeros((4 * n + 16, 4), dtype=np.float32) for k in names}
    cnt = {k: C.c_int(0) for k in names}
    args = []
    for k in names:
        args += [_p(bufs[k], c_fp), C.byref(cnt[k])]
    lib().orc_sr_extract(_p(xyz, c_fp), n, *args)
    return {k: bufs[k][:cnt[k].value].copy() for k in names}


def voxel_grid(pts, leaf):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.zeros((max(len(pts), 1), 4), dtype=np.float32)
    no = C.c_int(0)
    rc = lib().orc_voxel_grid(_p(pts, c_fp), len(pts), C.c_float(leaf), _p(out, c_fp), C.byref(no))
    return out[:no.value].copy(), rc


def loam_run(sweeps, stamps):
    """The four LOAM nodes in lock step over a list of raw sweeps [n,3] float32."""
    ns = len(sweeps)
    off = np.zeros(ns + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(a) for a in sweeps])
    xyz = np.ascontiguousarray(np.concatenate(sweeps), dtype=np.float32)
    stamps = np.ascontiguousarray(stamps, dtype=np.float64)
    lo = np.zeros((ns, 6), dtype=np.float32)
    lm = np.zeros((ns, 6), dtype=np.float32)
    tm = np.zeros((ns, 6), dtype=np.float32)
    track = np.zeros((ns, 4), dtype=np.float64)
    iters = np.zeros(ns, dtype=np.int32)
    lib().orc_loam_run(_p(xyz, c_fp), _p(off, C.POINTER(C.c_int)), ns, _p(stamps, c_dp), _p(lo, c_fp), _p(lm, c_fp),
                       _p(tm, c_fp), _p(track, c_dp), _p(iters, C.POINTER(C.c_int)))
    return {"lo_sum": lo, "lm_aft": lm, "tm_mapped": tm, "track": track, "lm_iters": iters}


def input_data_pass(sweeps, stamps, slam_distance, overlap):
    """input_data's replay + segmentation around the LOAM chain (one bag, one pass)."""
    ns = len(sweeps)
    off = np.zeros(ns + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(a) for a in sweeps])
    xyz = np.ascontiguousarray(np.concatenate(sweeps), dtype=np.float32)
    stamps = np.ascontiguousarray(stamps, dtype=np.float64)
    cap_t, cap_r = ns + 4, 4 * ns + 16
    first = np.zeros(cap_t, dtype=np.int32)
    last = np.zeros(cap_t, dtype=np.int32)
    toff = np.zeros(cap_t + 1, dtype=np.int32)
    rows = np.zeros((cap_r, 4), dtype=np.float64)
    ip = C.POINTER(C.c_int)
    nt = lib().orc_input_data_pass(_p(xyz, c_fp), _p(off, ip), ns, _p(stamps, c_dp), C.c_double(slam_distance),
                                   C.c_double(overlap), cap_t, _p(first, ip), _p(last, ip), _p(toff, ip),
                                   _p(rows, c_dp), cap_r)
    assert nt >= 0
    return [{"first": int(first[k]), "last": int(last[k]), "track": rows[toff[k]:toff[k + 1]].copy()} for k in range(nt)]


def parse_gps_log(text, t0, t1):
    """getGPS's dispatch on the sentence of the first line (GPRMC / GPGGA / GPGLL)."""
    if isinstance(text, str):
        text = text.encode()
    cap = text.count(b"\n") + 2
    lat, lon, t = np.empty(cap), np.empty(cap), np.empty(cap)
    n = lib().orc_parse_gps_log(text, C.c_size_t(len(text)), C.c_double(t0), C.c_double(t1),
                                _p(lat, c_dp), _p(lon, c_dp), _p(t, c_dp), cap)
    assert n >= 0
    return lat[:n].copy(), lon[:n].copy(), t[:n].copy()


def mars(lonlat, which):
    ll = f64(lonlat)
    out = np.empty_like(ll)
    getattr(lib(), "orc_" + which)(_p(ll, c_dp), len(ll), _p(out, c_dp))
    return out


def json_map(lonlat, flag, seg_end=None, rgb=None):
    ll = f64(lonlat)
    nseg = 0 if seg_end is None else len(seg_end)
    se = None if seg_end is None else np.ascontiguousarray(seg_end, dtype=np.int32)
    cc = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint32)
    lib().orc_json.restype = C.c_long
    need = lib().orc_json(None, C.c_size_t(0), _p(ll, c_dp), len(ll), flag, _p(se, c_ip), _p(cc, c_up), nseg)
    buf = C.create_string_buffer(need + 1)
    lib().orc_json(buf, C.c_size_t(need + 1), _p(ll, c_dp), len(ll), flag, _p(se, c_ip), _p(cc, c_up), nseg)
    return buf.value.decode()
