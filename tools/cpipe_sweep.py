#!/usr/bin/env python3
"""dctz_compress of a 1 GiB fp64 array (device entropy stage, tree-order mean) under the knobs of the pipelined path:
   python3 tools/cpipe_sweep.py "GROUP,MM_THREADS,FOLLOW_THREADS" ...      (0 = the default; "serial" = DCTZ_PIPELINE=0)
Prints the best of three calls per setting and the stage spans of that call."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from e2e_bench import TVar, StageTimes  # noqa: E402
import numpy as np  # noqa: E402
from tests import workloads as W  # noqa: E402

os.environ["DCTZ_QUIET"] = "1"
os.environ["DCTZ_ZLIB_GPU"] = "1"
os.environ["DCTZ_FAST_MEAN"] = "1"
lib = C.CDLL(os.path.join(ROOT, "dctz_amd", "lib", "libdctz-ec.so"))
lib.dctz_compress.argtypes = [C.POINTER(TVar), C.c_int, C.POINTER(C.c_size_t), C.POINTER(TVar), C.c_double]
lib.dctz_last_stage_times.argtypes = [C.POINTER(StageTimes)]
x0 = W.c3(512, seed=512, dtype=np.float64)
n = x0.size
zbuf = np.zeros(n * 8 + 4096, np.uint8)
for setting in sys.argv[1:]:
    for k in ("DCTZ_PIPELINE", "DCTZ_PIPE_GROUP", "DCTZ_PIPE_MM_THREADS", "DCTZ_PIPE_FOLLOW_THREADS"):
        os.environ.pop(k, None)
    if setting == "serial":
        os.environ["DCTZ_PIPELINE"] = "0"
    else:
        g, tm, tf = (int(v) for v in setting.split(","))
        if g:
            os.environ["DCTZ_PIPE_GROUP"] = str(g)
        if tm:
            os.environ["DCTZ_PIPE_MM_THREADS"] = str(tm)
        if tf:
            os.environ["DCTZ_PIPE_FOLLOW_THREADS"] = str(tf)
    best = None
    for rep in range(4):
        x = x0.copy()
        var, var_z = TVar(), TVar()
        var.datatype = var_z.datatype = 1
        var.buf.d = x.ctypes.data_as(C.POINTER(C.c_double))
        var_z.buf.d = zbuf.ctypes.data_as(C.POINTER(C.c_double))
        out = C.c_size_t(0)
        t0 = time.perf_counter()
        lib.dctz_compress(C.byref(var), n, C.byref(out), C.byref(var_z), 1e-3)
        t = time.perf_counter() - t0
        st = StageTimes()
        lib.dctz_last_stage_times(C.byref(st))
        if rep and (best is None or t < best[0]):
            best = (t, {k: round(getattr(st, k) * 1e3, 2) for k, _ in StageTimes._fields_})
    print(f"{setting:>24s}  {best[0] * 1e3:7.2f} ms  {best[1]}", flush=True)
