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
