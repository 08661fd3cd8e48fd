"""gpscalibration_amd -- MI355X-native scan-matching + GPS/SLAM track alignment.

Host-side Python mirror of include/gpscal.h (ctypes over libgpscal_hip.so).  The
arithmetic lives in hand-written HIP kernels (gpscalibration_amd/csrc); there is
no CPU fallback: importing works anywhere, creating a Context needs a gfx950 GPU.
"""
from ._lib import GpscalError, lib_path, load  # noqa: F401
from .api import Context, KnnIndex, ScanBatch  # noqa: F401

__all__ = ["Context", "KnnIndex", "ScanBatch", "GpscalError", "load", "lib_path"]
__version__ = "0.1"
