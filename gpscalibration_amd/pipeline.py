"""ctypes binding of libgpscal_host.so's in-process track pipeline (host C++ mirror of the
reference's long / short track nodes + KML writer, GPU arithmetic behind the C ABI)."""
import ctypes as C
import os

import numpy as np

from ._lib import load as _load_hip

_HERE = os.path.dirname(os.path.abspath(__file__))
_H = None


def load_host():
    global _H
    if _H is None:
        _load_hip()  # libgpscal_hip.so first (and torch's HIP runtime before it)
        path = os.path.join(_HERE, "libgpscal_host.so")
        if not os.path.exists(path):
            raise ImportError("libgpscal_host.so is missing: make -C gpscalibration_amd/host")
        _H = C.CDLL(path)
        _H.gpscal_host_pipeline.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_void_p,
                                            C.c_void_p]
        _H.gpscal_host_pipeline_sweeps.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_double, C.c_double, C.c_double, C.c_char_p, C.c_int, C.c_char_p,
                                                   C.c_char_p, C.c_void_p, C.c_void_p]
        _H.gpscal_host_read_bag.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p]
    return _H


def _pack(segs):
    off = np.zeros(len(segs) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(s) for s in segs])
    return np.ascontiguousarray(np.concatenate(segs), dtype=np.float64), off


def run_tracks(gps_log_path, longs, shorts, method="UTM", band_type=3, kml_original="", kml_calibrated=""):
    """Long pass -> short pass -> WGS84 + KML.  Returns dict(seconds=[long, short, output, total], points=[gps, calibrated])."""
    H = load_host()
    lx, lo = _pack(longs)
    sx, so = _pack(shorts)
    sec = np.zeros(4)
    npts = np.zeros(2, dtype=np.int32)
    rc = H.gpscal_host_pipeline(gps_log_path.encode(), lx.ctypes.data, lo.ctypes.data, len(longs), sx.ctypes.data,
                                so.ctypes.data, len(shorts), method.encode(), band_type, kml_original.encode(),
                                kml_calibrated.encode(), sec.ctypes.data, npts.ctypes.data)
    if rc != 0:
        raise RuntimeError("gpscal_host_pipeline failed (%d)" % rc)
    return {"seconds": sec.tolist(), "points": npts.tolist()}


def run_sweeps(gps_log_path, bags, stamps, long_distance, short_distance, overlap_distance, method="UTM",
               band_type=3, kml_original="", kml_calibrated=""):
    """Raw sweeps -> KML: input_data's replay + segmentation with the LOAM nodes on the GPU, then the
    track pipeline.  `bags` = list of lists of [n,3] float32 sweeps.  Returns dict(seconds=[slam, long,
    short, output, total], counts=[long tracks, short tracks, gps points, calibrated points])."""
    H = load_host()
    flat = [sw for b in bags for sw in b]
    bag_off = np.zeros(len(bags) + 1, dtype=np.int32)
    bag_off[1:] = np.cumsum([len(b) for b in bags])
    off = np.zeros(len(flat) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(a) for a in flat])
    xyz = np.ascontiguousarray(np.concatenate(flat), dtype=np.float32)
    st = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float64) for x in stamps]))
    sec = np.zeros(5)
    cnt = np.zeros(4, dtype=np.int32)
    rc = H.gpscal_host_pipeline_sweeps(gps_log_path.encode(), len(bags), xyz.ctypes.data, off.ctypes.data,
                                       bag_off.ctypes.data, st.ctypes.data, float(long_distance), float(short_distance),
                                       float(overlap_distance), method.encode(), band_type, kml_original.encode(),
                                       kml_calibrated.encode(), sec.ctypes.data, cnt.ctypes.data)
    if rc != 0:
        raise RuntimeError("gpscal_host_pipeline_sweeps failed (%d)" % rc)
    return {"seconds": sec.tolist(), "counts": cnt.tolist()}


def read_bag(path, topic="velodyne_points"):
    """The PointCloud2 messages of one rosbag V2.0 file -> (list of [n,3] float32 sweeps, stamps)."""
    H = load_host()
    nm, npt = C.c_int(0), C.c_int(0)
    dummy = np.zeros(1, dtype=np.float32)
    rc = H.gpscal_host_read_bag(path.encode(), topic.encode(), dummy.ctypes.data, 0, dummy.ctypes.data,
                                dummy.ctypes.data, 0, C.byref(nm), C.byref(npt))
    if rc == 1:
        raise RuntimeError("cannot read %s" % path)
    xyz = np.zeros((max(npt.value, 1), 3), dtype=np.float32)
    off = np.zeros(nm.value + 1, dtype=np.int32)
    st = np.zeros(max(nm.value, 1), dtype=np.float64)
    rc = H.gpscal_host_read_bag(path.encode(), topic.encode(), xyz.ctypes.data, len(xyz), off.ctypes.data,
                                st.ctypes.data, nm.value, C.byref(nm), C.byref(npt))
    if rc != 0:
        raise RuntimeError("cannot read %s (%d)" % (path, rc))
    return [xyz[off[k]:off[k + 1]].copy() for k in range(nm.value)], st[:nm.value].copy()
