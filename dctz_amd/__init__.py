"""dctz_amd -- MI355X (gfx950) implementation of DCTZ's block-DCT + binning hot path.

The product is the C-ABI library lib/libdctzhip.so (include/dctz_hip.h) and the
drop-in host libraries lib/libdctz-{ec,qt}.so (include/dctz.h).  This package is
only the thin Python plumbing the tests and bench.py use to drive them; it has
no compute path of its own and raises if the HIP library is missing.
"""
from .hip import (EC, F32, F64, QT, CompressInfo, Context, DctzHipError, lib_path,
                  load_library)

__all__ = ["Context", "CompressInfo", "DctzHipError", "load_library", "lib_path",
           "EC", "QT", "F32", "F64"]
__version__ = "0.1.0"
