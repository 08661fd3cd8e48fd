"""Python host mirror of include/gpscal.h.

Every array argument may be a numpy array (host memory) or a torch CUDA tensor
(device memory, used in place by the kernels): the C ABI takes plain pointers and
asks the HIP runtime which side they live on.  PyTorch is plumbing only.

Names follow the reference's interface for the path:
  Context.weights_speed / weights_irls   <- WeightCoeCal::ICPWeightCoeCal (weight_calculation.h:14-16)
  Context.track_fit                      <- trackCalibration(...), doICP(), doCalibration() (track_calibration.h:12-23)
  Context.long_segment                   <- longDisTrackPro body (long_distance_track_process.cpp:58-83)
  Context.wgs_to_enu / enu_to_wgs / gps_to_enu <- GPSPro (gps_process.h:53-56, 73-76)
  KnnIndex.search                        <- pcl::KdTreeFLANN::nearestKSearch (laserOdometry.cpp:603, laserMapping.cpp:760)
  ScanBatch.icp                          <- the generic scan-matching iteration
"""
import ctypes as C

import numpy as np

from ._lib import GpscalError, load

METHOD = {"UTM": 0, "Gaussion": 1, "Gauss": 1, 0: 0, 1: 1}


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class _DevPtr(C.c_void_p):
    """The address of a CUDA tensor: _Ordered.call orders the library's stream against torch's for calls that get one."""


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        assert x.is_contiguous()
        return _DevPtr(x.data_ptr()) if x.is_cuda else C.c_void_p(x.data_ptr())
    assert x.flags["C_CONTIGUOUS"]
    return C.c_void_p(x.ctypes.data)


def _f64(x):
    if x is None or _is_torch(x):
        return x
    return np.ascontiguousarray(x, dtype=np.float64)


def _f32(x):
    if x is None or _is_torch(x):
        return x
    return np.ascontiguousarray(x, dtype=np.float32)


class _Ordered:
    """The C ABI with stream ordering for torch tensors (include/gpscal.h, conventions): a call that was handed a
    CUDA tensor first makes the context's stream wait for torch's current stream (the tensor may still be being
    written by a pending torch kernel) and afterwards makes torch's current stream wait for the context's stream
    (a tensor the library writes is then safe to use from torch without a host synchronisation)."""

    def __init__(self, L, ctx):
        self._L, self._ctx = L, ctx

    def __getattr__(self, name):
        fn = getattr(self._L, name)

        def call(*args):
            # (the marker travels with the argument list: no state shared between threads or left behind by a
            # failed call)
            if not any(isinstance(a, _DevPtr) for a in args):
                return fn(*args)
            import torch
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self._L.gpscal_wait_for_stream(self._ctx._h, stream)
            rc = fn(*args)
            self._L.gpscal_make_stream_wait(self._ctx._h, stream)
            return rc

        return call


class Context:
    """One GPU, one HIP stream (gpscal_ctx).  Fails loudly without a gfx950 device."""

    def __init__(self, device_id=0):
        L = load()
        h = C.c_void_p()
        rc = L.gpscal_create(C.byref(h), int(device_id), 0)
        if rc:
            raise GpscalError(rc, "gpscal_create(device %d)" % device_id)
        self._h = h
        self._L = _Ordered(L, self)
        self.device_id = device_id
        self._children = []  # weak references to the scan batches and k-NN indexes built on this context

    def _adopt(self, child):
        import weakref
        self._children = [r for r in self._children if r() is not None]
        self._children.append(weakref.ref(child))
        return child

    def close(self):
        if getattr(self, "_h", None):
            # whatever still lives on this context goes first: its device blocks belong to the context's stream cache
            for r in getattr(self, "_children", []):
                c = r()
                if c is not None:
                    c.close()
            self._children = []
            self._L.gpscal_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _ck(self, rc, what):
        if rc:
            raise GpscalError(rc, "%s: %s" % (what, self._L.gpscal_last_error(self._h).decode()))

    def sync(self):
        self._ck(self._L.gpscal_sync(self._h), "gpscal_sync")

    @property
    def stream(self):
        return self._L.gpscal_stream(self._h)

    def info(self):
        buf = C.create_string_buffer(256)
        self._L.gpscal_device_info(self._h, buf, 256)
        return buf.value.decode()

    # ------------------------------------------------------------ weights
    def weights_speed(self, slam_xyzt):
        s = _f64(slam_xyzt)
        n = len(s)
        w = np.empty(n)
        self._ck(self._L.gpscal_weights_speed(self._h, _ptr(s), n, _ptr(w)), "weights_speed")
        return w

    def weights_irls(self, slam_xyzt, enu_xyzt, fit_xyzt):
        s, e, f = _f64(slam_xyzt), _f64(enu_xyzt), _f64(fit_xyzt)
        n = len(s)
        if len(e) != n or len(f) != n:
            raise GpscalError(-5, "weights_irls: sizes differ")
        w = np.empty(n)
        self._ck(self._L.gpscal_weights_irls(self._h, _ptr(s), _ptr(e), _ptr(f), n, _ptr(w)), "weights_irls")
        return w

    # -------------------------------------------------------------- track
    def track_fit(self, slam_xyzt, enu_xyzt, w, seg_offsets=None):
        """Returns (T, rotated_xyz, calibrated_xyzt).  With seg_offsets, T is (nseg,4,4)."""
        s, e, w = _f64(slam_xyzt), _f64(enu_xyzt), _f64(w)
        n = len(s)
        if len(e) != n or len(w) != n:  # the reference exit(1)s here (track_calibration.cc:46-50)
            raise GpscalError(-5, "track_fit: SLAM/ENU/weight sizes differ")
        rot = np.empty((n, 3))
        cal = np.empty((n, 4))
        if seg_offsets is None:
            T = np.empty((4, 4))
            self._ck(self._L.gpscal_track_fit(self._h, _ptr(s), _ptr(e), _ptr(w), n, _ptr(T), _ptr(rot), _ptr(cal)),
                     "track_fit")
        else:
            so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
            nseg = len(so) - 1
            T = np.empty((nseg, 4, 4))
            self._ck(self._L.gpscal_track_fit_batched(self._h, _ptr(s), _ptr(e), _ptr(w), _ptr(so), nseg, _ptr(T),
                                                      _ptr(rot), _ptr(cal)), "track_fit_batched")
        return T, rot, cal

    def long_segment(self, slam_xyzt, enu_xyzt, irls_iters=5, seg_offsets=None):
        """Returns (w_final, last_fit_xyzt)."""
        s, e = _f64(slam_xyzt), _f64(enu_xyzt)
        n = len(s)
        if len(e) != n:
            raise GpscalError(-5, "long_segment: SLAM/ENU sizes differ")
        w = np.empty(n)
        fit = np.empty((n, 4))
        if seg_offsets is None:
            self._ck(self._L.gpscal_long_segment(self._h, _ptr(s), _ptr(e), n, irls_iters, _ptr(w), _ptr(fit)),
                     "long_segment")
        else:
            so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
            self._ck(self._L.gpscal_long_segment_batched(self._h, _ptr(s), _ptr(e), _ptr(so), len(so) - 1,
                                                         irls_iters, _ptr(w), _ptr(fit)), "long_segment_batched")
        return w, fit

    # ---------------------------------------------------------------- geo
    def wgs_to_enu(self, lat, lon, method="UTM", band_type=3):
        lat, lon = _f64(lat), _f64(lon)
        xy = np.empty((len(lat), 2))
        self._ck(self._L.gpscal_wgs_to_enu(self._h, METHOD[method], band_type, _ptr(lat), _ptr(lon), len(lat),
                                           _ptr(xy)), "wgs_to_enu")
        return xy

    def enu_to_wgs(self, enu_xyztw, method="UTM", band_type=3):
        e = _f64(enu_xyztw)
        ll = np.empty((len(e), 2))
        alt = np.empty(len(e))
        self._ck(self._L.gpscal_enu_to_wgs(self._h, METHOD[method], band_type, _ptr(e), len(e), _ptr(ll), _ptr(alt)),
                 "enu_to_wgs")
        return ll, alt

    def gps_to_enu(self, lat, lon, gps_t, slam_xyzt, method="UTM", band_type=3):
        lat, lon, gt, s = _f64(lat), _f64(lon), _f64(gps_t), _f64(slam_xyzt)
        enu = np.empty((len(s), 4))
        k = C.c_int(0)
        self._ck(self._L.gpscal_gps_to_enu(self._h, METHOD[method], band_type, _ptr(lat), _ptr(lon), _ptr(gt),
                                           len(gt), _ptr(s), len(s), _ptr(enu), C.byref(k)), "gps_to_enu")
        return enu[:k.value].copy()

    def height_compensate(self, loam_xyzt):
        p = _f64(loam_xyzt)
        out = np.empty((len(p), 4))
        self._ck(self._L.gpscal_height_compensate(self._h, _ptr(p), len(p), _ptr(out)), "height_compensate")
        return out

    # --------------------------------------------------------------- LOAM
    def loam_odometry(self, sharp, flat, corner_last, surf_last, transform_in=None, transform_sum_in=None):
        """Batched laserOdometry iteration loop.  Each cloud argument is a list of [n,4] float32 arrays
        (one per sweep).  Returns (transform[nsweeps,6], iters, nsel, transform_sum | None)."""
        ns = len(sharp)

        def pack(lst):
            off = np.zeros(ns + 1, dtype=np.int32)
            off[1:] = np.cumsum([len(a) for a in lst])
            data = np.ascontiguousarray(np.concatenate(lst) if off[-1] else np.zeros((1, 4)), dtype=np.float32)
            return data, off
        sh, sho = pack(sharp)
        fl, flo = pack(flat)
        cl, clo = pack(corner_last)
        sl, slo = pack(surf_last)
        tin = np.zeros((ns, 6), dtype=np.float32) if transform_in is None else np.ascontiguousarray(transform_in, dtype=np.float32)
        tout = np.empty((ns, 6), dtype=np.float32)
        iters = np.empty(ns, dtype=np.int32)
        nsel = np.empty(ns, dtype=np.int32)
        sin = None if transform_sum_in is None else np.ascontiguousarray(transform_sum_in, dtype=np.float32)
        sout = None if sin is None else np.empty((ns, 6), dtype=np.float32)
        self._ck(self._L.gpscal_loam_odometry_batched(self._h, ns, _ptr(sh), _ptr(sho), _ptr(fl), _ptr(flo), _ptr(cl),
                                                      _ptr(clo), _ptr(sl), _ptr(slo), _ptr(tin), _ptr(tout), _ptr(iters),
                                                      _ptr(nsel), _ptr(sin), _ptr(sout)), "loam_odometry_batched")
        return tout, iters, nsel, sout

    def loam_mapping(self, corner_stack, surf_stack, corner_map, surf_map, transform_in=None):
        """Batched laserMapping optimisation loop (laserMapping.cpp:748-1018).  Each cloud argument is a
        list of [n,4] float32 arrays (one per sweep).  Returns (transformTobeMapped[nsweeps,6], iters, nsel)."""
        ns = len(corner_stack)

        def pack(lst):
            off = np.zeros(ns + 1, dtype=np.int32)
            off[1:] = np.cumsum([len(a) for a in lst])
            data = np.ascontiguousarray(np.concatenate(lst) if off[-1] else np.zeros((1, 4)), dtype=np.float32)
            return data, off
        cs, cso = pack(corner_stack)
        ss, sso = pack(surf_stack)
        cm, cmo = pack(corner_map)
        sm, smo = pack(surf_map)
        tin = np.zeros((ns, 6), dtype=np.float32) if transform_in is None else np.ascontiguousarray(transform_in, dtype=np.float32)
        tout = np.empty((ns, 6), dtype=np.float32)
        iters = np.empty(ns, dtype=np.int32)
        nsel = np.empty(ns, dtype=np.int32)
        self._ck(self._L.gpscal_loam_mapping_batched(self._h, ns, _ptr(cs), _ptr(cso), _ptr(ss), _ptr(sso), _ptr(cm),
                                                     _ptr(cmo), _ptr(sm), _ptr(smo), _ptr(tin), _ptr(tout), _ptr(iters),
                                                     _ptr(nsel)), "loam_mapping_batched")
        return tout, iters, nsel

    def scan_registration(self, sweeps, less_flat_factor=2):
        """Batched scanRegistration (scanRegistration.cpp:238-674).  `sweeps` = list of [n,3] float32 raw
        sweeps.  Returns a list of dict(full, sharp, less_sharp, flat, less_flat) of [k,4] float32."""
        ns = len(sweeps)
        off = np.zeros(ns + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(a) for a in sweeps])
        lfo = (off * less_flat_factor).astype(np.int32)
        xyz = np.ascontiguousarray(np.concatenate(sweeps) if off[-1] else np.zeros((1, 3)), dtype=np.float32)
        tot = max(int(off[-1]), 1)
        full = np.empty((tot, 4), dtype=np.float32)
        sharp = np.empty((ns, 1536, 4), dtype=np.float32)
        lsharp = np.empty((ns, 1920, 4), dtype=np.float32)
        flat = np.empty((ns, 3072, 4), dtype=np.float32)
        lflat = np.empty((max(int(lfo[-1]), 1), 4), dtype=np.float32)
        cnt = np.empty((ns, 5), dtype=np.int32)
        self._ck(self._L.gpscal_scan_registration_batched(self._h, ns, _ptr(xyz), _ptr(off), _ptr(full), _ptr(sharp),
                                                          _ptr(lsharp), _ptr(flat), _ptr(lflat), _ptr(lfo), _ptr(cnt)),
                 "scan_registration_batched")
        out = []
        for b in range(ns):
            out.append({"full": full[off[b]:off[b] + cnt[b, 0]].copy(), "sharp": sharp[b, :cnt[b, 1]].copy(),
                        "less_sharp": lsharp[b, :cnt[b, 2]].copy(), "flat": flat[b, :cnt[b, 3]].copy(),
                        "less_flat": lflat[lfo[b]:lfo[b] + cnt[b, 4]].copy()})
        return out

    def voxel_grid(self, clouds, leaf):
        """Batched pcl::VoxelGrid with a cubic leaf.  `clouds` = list of [n,4] float32; returns a list."""
        nc = len(clouds)
        off = np.zeros(nc + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(a) for a in clouds])
        pts = np.ascontiguousarray(np.concatenate(clouds) if off[-1] else np.zeros((1, 4)), dtype=np.float32)
        out = np.empty_like(pts)
        cnt = np.empty(nc, dtype=np.int32)
        self._ck(self._L.gpscal_voxel_grid_batched(self._h, nc, _ptr(pts), _ptr(off), float(leaf), _ptr(out), _ptr(cnt)),
                 "voxel_grid_batched")
        return [out[off[c]:off[c] + cnt[c]].copy() for c in range(nc)]

    @staticmethod
    def loam_pack(segments, stamps):
        """The argument arrays of gpscal_loam_run_batched for `segments` = list (one per segment) of lists of raw
        sweeps [n,3] float32 and `stamps` = list of per-segment stamp arrays: (xyz, off, seg_off, stamps)."""
        nseg = len(segments)
        flat = [sw for seg in segments for sw in seg]
        nsw = len(flat)
        seg_off = np.zeros(nseg + 1, dtype=np.int32)
        seg_off[1:] = np.cumsum([len(seg) for seg in segments])
        off = np.zeros(nsw + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(a) for a in flat])
        xyz = np.ascontiguousarray(np.concatenate(flat), dtype=np.float32)
        st = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float64) for x in stamps]))
        return xyz, off, seg_off, st

    def loam_run_packed(self, packed, corner_pool_cap=0, surf_pool_cap=0):
        """gpscal_loam_run_batched on arrays from loam_pack; `xyz` may be a torch tensor in HBM (used in place)."""
        xyz, off, seg_off, st = packed
        nseg, nsw = len(seg_off) - 1, len(off) - 1
        lo = np.empty((nsw, 6), dtype=np.float32)
        lm = np.empty((nsw, 6), dtype=np.float32)
        tm = np.empty((nsw, 6), dtype=np.float32)
        track = np.empty((nsw, 4), dtype=np.float64)
        iters = np.empty(nsw, dtype=np.int32)
        self._ck(self._L.gpscal_loam_run_batched(self._h, nseg, _ptr(xyz), _ptr(off), _ptr(seg_off), _ptr(st), _ptr(lo),
                                                 _ptr(lm), _ptr(tm), _ptr(track), _ptr(iters), int(corner_pool_cap),
                                                 int(surf_pool_cap)), "loam_run_batched")
        out = []
        for s in range(nseg):
            a, b = seg_off[s], seg_off[s + 1]
            out.append({"lo_sum": lo[a:b], "lm_aft": lm[a:b], "tm_mapped": tm[a:b], "track": track[a:b],
                        "lm_iters": iters[a:b]})
        return out

    def loam_run(self, segments, stamps, corner_pool_cap=0, surf_pool_cap=0):
        """The four LOAM nodes over pre-cut segments.  `segments` = list (one per segment) of lists of raw
        sweeps [n,3] float32; `stamps` = list of per-segment stamp arrays.  Returns a list of dicts
        (lo_sum, lm_aft, tm_mapped [n,6] float32; track [n,4] float64; lm_iters [n] int32) per segment."""
        return self.loam_run_packed(self.loam_pack(segments, stamps), corner_pool_cap, surf_pool_cap)

    def input_data_run(self, bags, stamps, long_distance, short_distance, overlap_distance, corner_pool_cap=0,
                       surf_pool_cap=0):
        """input_data's replay + segmentation around the LOAM chain (input_data.cpp:78-124, 266-444).
        `bags` = list of lists of raw sweeps; returns a list of dict(flag, bag, first, last, track[n,4])."""
        nbag = len(bags)
        flat = [sw for b in bags for sw in b]
        nsw = len(flat)
        bag_off = np.zeros(nbag + 1, dtype=np.int32)
        bag_off[1:] = np.cumsum([len(b) for b in bags])
        off = np.zeros(nsw + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(a) for a in flat])
        xyz = np.ascontiguousarray(np.concatenate(flat), dtype=np.float32)
        st = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float64) for x in stamps]))
        cap_t, cap_r = 2 * nsw + 8, 8 * nsw + 16
        flag = np.zeros(cap_t, dtype=np.int32)
        bag = np.zeros(cap_t, dtype=np.int32)
        first = np.zeros(cap_t, dtype=np.int32)
        last = np.zeros(cap_t, dtype=np.int32)
        toff = np.zeros(cap_t + 1, dtype=np.int32)
        rows = np.zeros((cap_r, 4), dtype=np.float64)
        nt = np.zeros(1, dtype=np.int32)
        self._ck(self._L.gpscal_input_data_run(self._h, nbag, _ptr(xyz), _ptr(off), _ptr(bag_off), _ptr(st),
                                               float(long_distance), float(short_distance), float(overlap_distance),
                                               cap_t, _ptr(flag), _ptr(bag), _ptr(first), _ptr(last), _ptr(toff),
                                               _ptr(rows), cap_r, _ptr(nt), int(corner_pool_cap), int(surf_pool_cap)),
                 "input_data_run")
        return [{"flag": int(flag[k]), "bag": int(bag[k]), "first": int(first[k]), "last": int(last[k]),
                 "track": rows[toff[k]:toff[k + 1]].copy()} for k in range(int(nt[0]))]

    # ------------------------------------------------------------ RCCL (exported exchange of SURVEY 8e)
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id; rank 0 creates it and hands it to the other ranks out of band."""
        buf = (C.c_char * 128)()
        rc = load().gpscal_comm_unique_id(buf)
        if rc:
            raise GpscalError(rc, "gpscal_comm_unique_id")
        return bytes(buf)

    def comm_init(self, uid, rank, world):
        buf = (C.c_char * 128).from_buffer_copy(bytes(uid))
        self._ck(self._L.gpscal_comm_init(self._h, buf, int(rank), int(world)), "gpscal_comm_init")

    def comm_destroy(self):
        self._ck(self._L.gpscal_comm_destroy(self._h), "gpscal_comm_destroy")

    def allgather_chains(self, local, counts):
        """Ragged all-gather of float64 values over RCCL (gpscal_allgather_chains): `local` holds counts[rank]
        doubles, every rank gets all sum(counts) in rank order."""
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1)
        out = np.empty(int(counts.sum()), dtype=np.float64)
        self._ck(self._L.gpscal_allgather_chains(self._h, _ptr(local) if len(local) else None, _ptr(counts), _ptr(out)),
                 "gpscal_allgather_chains")
        return out

    def imgps_message(self, calibrated_xyztw, method="UTM", band_type=3):
        """/imorpheus_gps payload (result_control 4, short_distance_track_process.cpp:295-309): [n,3] {b, l, w}."""
        e = _f64(calibrated_xyztw)
        out = np.empty((len(e), 3))
        self._ck(self._L.gpscal_imgps_message(self._h, METHOD[method], band_type, _ptr(e), len(e), _ptr(out)), "imgps_message")
        return out

    def mars(self, lonlat, which):
        """GCJ-02 / BD-09 conversions of [n,2] {lon, lat}: which = "gps_to_gcj" | "gcj_to_bd" | "bd_to_gcj"."""
        p = _f64(lonlat)
        out = np.empty_like(p)
        self._ck(getattr(self._L, "gpscal_" + which)(self._h, _ptr(p), len(p), _ptr(out)), which)
        return out

    def loam_transform(self, transform6, pts_xyzi, to_end=False):
        t = np.ascontiguousarray(transform6, dtype=np.float32)
        p = np.ascontiguousarray(pts_xyzi, dtype=np.float32)
        out = np.empty_like(p)
        self._ck(self._L.gpscal_loam_transform(self._h, _ptr(t), _ptr(p), len(p), _ptr(out), int(to_end)), "loam_transform")
        return out

    # ------------------------------------------------------------ factories
    def knn_index(self, xyz, stride_bytes=12, cell_size=0.0):
        return self._adopt(KnnIndex(self, xyz, stride_bytes, cell_size))

    def scan_batch(self, tgt_xyz, tgt_off, src_xyz, src_off, w=None, cell_size=0.0):
        return self._adopt(ScanBatch(self, tgt_xyz, tgt_off, src_xyz, src_off, w, cell_size))


class KnnIndex:
    """Exact k-NN index over one cloud (gpscal_knn_build / _search / _free)."""

    def __init__(self, ctx, xyz, stride_bytes=12, cell_size=0.0):
        self.ctx = ctx
        self._xyz = _f32(xyz)
        if _is_torch(self._xyz):
            m = self._xyz.numel() * self._xyz.element_size() // stride_bytes
        else:
            m = self._xyz.size * 4 // stride_bytes
        self.m = m
        h = C.c_void_p()
        ctx._ck(ctx._L.gpscal_knn_build(ctx._h, _ptr(self._xyz), m, stride_bytes, C.c_float(cell_size), C.byref(h)),
                "knn_build")
        self._h = h

    def search(self, query_xyz, k=1, stride_bytes=12, out_idx=None, out_sqd=None):
        q = _f32(query_xyz)
        n = (q.numel() * q.element_size() if _is_torch(q) else q.size * 4) // stride_bytes
        idx = out_idx if out_idx is not None else np.empty((n, k), dtype=np.int32)
        sqd = out_sqd if out_sqd is not None else np.empty((n, k), dtype=np.float32)
        self.ctx._ck(self.ctx._L.gpscal_knn_search(self._h, _ptr(q), n, stride_bytes, k, _ptr(idx), _ptr(sqd)),
                     "knn_search")
        return idx, sqd

    def icp_run(self, src_xyz, iters, T0=None, w=None, stride_bytes=12):
        s = _f32(src_xyz)
        n = (s.numel() * s.element_size() if _is_torch(s) else s.size * 4) // stride_bytes
        T0 = None if T0 is None else _f64(np.asarray(T0).reshape(16))
        w = _f64(w)
        T = np.empty((4, 4))
        hist = np.empty(max(iters, 1))
        self.ctx._ck(self.ctx._L.gpscal_icp_run(self.ctx._h, self._h, _ptr(s), n, stride_bytes, _ptr(w), iters,
                                                _ptr(T0), _ptr(T), _ptr(hist)), "icp_run")
        return T, hist[:iters]

    def icp_iterate(self, src_xyz, T_in, w=None, stride_bytes=12):
        T, hist = self.icp_run(src_xyz, 1, T_in, w, stride_bytes)
        return T, float(hist[0])

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._L.gpscal_knn_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class ScanBatch:
    """npairs independent (target, source) scan pairs resident in HBM."""

    def __init__(self, ctx, tgt_xyz, tgt_off, src_xyz, src_off, w=None, cell_size=0.0):
        self.ctx = ctx
        self._keep = (_f32(tgt_xyz), _f32(src_xyz), _f64(w))
        to = np.ascontiguousarray(tgt_off, dtype=np.int64)
        so = np.ascontiguousarray(src_off, dtype=np.int64)
        self.npairs = len(to) - 1
        self.n_total = int(so[-1] - so[0])
        self.m_total = int(to[-1] - to[0])
        h = C.c_void_p()
        ctx._ck(ctx._L.gpscal_scan_batch_create(ctx._h, self.npairs, _ptr(self._keep[0]), _ptr(to),
                                                _ptr(self._keep[1]), _ptr(so), _ptr(self._keep[2]),
                                                C.c_float(cell_size), C.byref(h)), "scan_batch_create")
        self._h = h

    def set_pose(self, T0=None):
        T0 = None if T0 is None else _f64(np.asarray(T0).reshape(self.npairs, 16))
        self.ctx._ck(self.ctx._L.gpscal_scan_batch_set_pose(self._h, _ptr(T0)), "scan_batch_set_pose")

    def icp(self, iters, want_err=True, profile=False, T_out=None, err_out=None):
        """Runs `iters` iterations.  Returns (T[npairs,4,4], mean_err[npairs,iters] | None, step_ms | None).
        `T_out` / `err_out` (float64, [npairs,4,4] / [npairs,iters]; host arrays or CUDA tensors) receive the poses and
        the error history in place: with device tensors the call returns after enqueue."""
        T = T_out if T_out is not None else np.empty((self.npairs, 4, 4))
        err = err_out if err_out is not None else (np.empty((self.npairs, iters)) if want_err else None)
        ms = np.empty(iters, dtype=np.float32) if profile else None
        self.ctx._ck(self.ctx._L.gpscal_scan_batch_icp(self._h, iters, _ptr(T), _ptr(err), _ptr(ms)),
                     "scan_batch_icp")
        return T, err, ms

    def correspondences(self):
        idx = np.empty(self.n_total, dtype=np.int32)
        sqd = np.empty(self.n_total, dtype=np.float32)
        self.ctx._ck(self.ctx._L.gpscal_scan_batch_correspondences(self._h, _ptr(idx), _ptr(sqd)),
                     "scan_batch_correspondences")
        return idx, sqd

    @property
    def build_seconds(self):
        return self.ctx._L.gpscal_scan_batch_build_seconds(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._L.gpscal_scan_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
