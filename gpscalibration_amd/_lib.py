"""Loads libgpscal_hip.so (in-tree build) and declares the C ABI of include/gpscal.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXPORTS = [
    "gpscal_create", "gpscal_destroy", "gpscal_sync", "gpscal_stream", "gpscal_wait_for_stream",
    "gpscal_make_stream_wait", "gpscal_strerror",
    "gpscal_last_error", "gpscal_device_info",
    "gpscal_weights_speed", "gpscal_weights_irls",
    "gpscal_track_fit", "gpscal_track_fit_batched", "gpscal_long_segment", "gpscal_long_segment_batched",
    "gpscal_wgs_to_enu", "gpscal_enu_to_wgs", "gpscal_gps_to_enu", "gpscal_gps_to_enu_batched",
    "gpscal_height_compensate",
    "gpscal_knn_build", "gpscal_knn_search", "gpscal_knn_free",
    "gpscal_scan_batch_create", "gpscal_scan_batch_set_pose", "gpscal_scan_batch_icp",
    "gpscal_scan_batch_correspondences", "gpscal_scan_batch_build_seconds", "gpscal_scan_batch_destroy",
    "gpscal_icp_iterate", "gpscal_icp_run",
    "gpscal_loam_odometry_batched", "gpscal_loam_mapping_batched", "gpscal_loam_transform",
    "gpscal_scan_registration_batched", "gpscal_voxel_grid_batched", "gpscal_loam_run_batched", "gpscal_input_data_run",
    "gpscal_gps_to_gcj", "gpscal_gcj_to_bd", "gpscal_bd_to_gcj", "gpscal_imgps_message",
    "gpscal_comm_unique_id", "gpscal_comm_init", "gpscal_allgather_chains", "gpscal_comm_destroy",
]


class GpscalError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__("gpscal error %d (%s)%s" % (code, _strerror(code), (": " + what) if what else ""))


def lib_path():
    # GPSCAL_LIB: another build of the library (tuning experiments)
    return os.environ.get("GPSCAL_LIB") or os.path.join(_HERE, "libgpscal_hip.so")


def _strerror(code):
    try:
        return load().gpscal_strerror(code).decode()
    except Exception:  # noqa: BLE001
        return "?"


def load():
    """Returns the ctypes handle; raises if the HIP extension has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "libgpscal_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gpscalibration_amd/csrc`.  There is no CPU fallback." % path)
    # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64 / librccl.  Two HIP
    # runtimes in one process do not coexist (the second sees no GPU), so when torch is
    # installed its copies must be the ones the process binds first.
    if not os.environ.get("GPSCAL_NO_TORCH"):  # debugging aid: bind the system HIP runtime instead
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(path)
    vp, i, dp, fp, ip = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p
    L.gpscal_strerror.restype = C.c_char_p
    L.gpscal_strerror.argtypes = [i]
    L.gpscal_last_error.restype = C.c_char_p
    L.gpscal_last_error.argtypes = [vp]
    L.gpscal_stream.restype = vp
    L.gpscal_stream.argtypes = [vp]
    L.gpscal_wait_for_stream.argtypes = [vp, vp]
    L.gpscal_make_stream_wait.argtypes = [vp, vp]
    L.gpscal_create.argtypes = [C.POINTER(vp), i, C.c_uint]
    L.gpscal_destroy.argtypes = [vp]
    L.gpscal_sync.argtypes = [vp]
    L.gpscal_device_info.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.gpscal_weights_speed.argtypes = [vp, dp, i, dp]
    L.gpscal_weights_irls.argtypes = [vp, dp, dp, dp, i, dp]
    L.gpscal_track_fit.argtypes = [vp, dp, dp, dp, i, dp, dp, dp]
    L.gpscal_track_fit_batched.argtypes = [vp, dp, dp, dp, ip, i, dp, dp, dp]
    L.gpscal_long_segment.argtypes = [vp, dp, dp, i, i, dp, dp]
    L.gpscal_long_segment_batched.argtypes = [vp, dp, dp, ip, i, i, dp, dp]
    L.gpscal_wgs_to_enu.argtypes = [vp, i, i, dp, dp, i, dp]
    L.gpscal_enu_to_wgs.argtypes = [vp, i, i, dp, i, dp, dp]
    L.gpscal_gps_to_enu.argtypes = [vp, i, i, dp, dp, dp, i, dp, i, dp, C.POINTER(i)]
    L.gpscal_gps_to_enu_batched.argtypes = [vp, i, i, dp, dp, dp, ip, dp, ip, i, dp, ip]
    L.gpscal_height_compensate.argtypes = [vp, dp, i, dp]
    L.gpscal_knn_build.argtypes = [vp, fp, i, i, C.c_float, C.POINTER(vp)]
    L.gpscal_knn_search.argtypes = [vp, fp, i, i, i, ip, fp]
    L.gpscal_knn_free.argtypes = [vp]
    L.gpscal_scan_batch_create.argtypes = [vp, i, fp, ip, fp, ip, dp, C.c_float, C.POINTER(vp)]
    L.gpscal_scan_batch_set_pose.argtypes = [vp, dp]
    L.gpscal_scan_batch_icp.argtypes = [vp, i, dp, dp, fp]
    L.gpscal_scan_batch_correspondences.argtypes = [vp, ip, fp]
    L.gpscal_scan_batch_build_seconds.restype = C.c_double
    L.gpscal_scan_batch_build_seconds.argtypes = [vp]
    L.gpscal_scan_batch_destroy.argtypes = [vp]
    L.gpscal_icp_iterate.argtypes = [vp, vp, fp, i, i, dp, dp, dp, dp]
    L.gpscal_icp_run.argtypes = [vp, vp, fp, i, i, dp, i, dp, dp, dp]
    L.gpscal_loam_odometry_batched.argtypes = [vp, i, fp, ip, fp, ip, fp, ip, fp, ip, fp, fp, ip, ip, fp, fp]
    L.gpscal_loam_mapping_batched.argtypes = [vp, i, fp, ip, fp, ip, fp, ip, fp, ip, fp, fp, ip, ip]
    L.gpscal_scan_registration_batched.argtypes = [vp, i, fp, ip, fp, fp, fp, fp, fp, ip, ip]
    L.gpscal_voxel_grid_batched.argtypes = [vp, i, fp, ip, C.c_float, fp, ip]
    L.gpscal_loam_run_batched.argtypes = [vp, i, fp, ip, ip, dp, fp, fp, fp, dp, ip, i, i]
    L.gpscal_input_data_run.argtypes = [vp, i, fp, ip, ip, dp, C.c_double, C.c_double, C.c_double, i, ip, ip, ip, ip, ip,
                                         dp, i, ip, i, i]
    for name in ("gpscal_gps_to_gcj", "gpscal_gcj_to_bd", "gpscal_bd_to_gcj"):
        getattr(L, name).argtypes = [vp, dp, i, dp]
    L.gpscal_imgps_message.argtypes = [vp, i, i, dp, i, dp]
    L.gpscal_loam_transform.argtypes = [vp, fp, fp, i, fp, i]
    L.gpscal_comm_unique_id.argtypes = [vp]
    L.gpscal_comm_init.argtypes = [vp, vp, i, i]
    L.gpscal_allgather_chains.argtypes = [vp, dp, ip, dp]
    L.gpscal_comm_destroy.argtypes = [vp]
    _LIB = L
    return L
