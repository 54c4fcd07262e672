#!/usr/bin/env python3
"""One GPU-tail compress + decompress of the 512^3 shard through the drop-in (development: DCTZ_PIPE_DEBUG=1 prints the
pipelined calls' timelines)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests import workloads as W
sys.path.insert(0, os.path.join(ROOT, "tools"))
from e2e_bench import TVar
os.environ["DCTZ_QUIET"] = "1"; os.environ["DCTZ_ZLIB_GPU"] = "1"; os.environ["DCTZ_FAST_MEAN"] = "1"
lib = C.CDLL(os.path.join(ROOT, "dctz_amd", "lib", "libdctz-ec.so"))
lib.dctz_compress.argtypes = [C.POINTER(TVar), C.c_int, C.POINTER(C.c_size_t), C.POINTER(TVar), C.c_double]
lib.dctz_decompress.argtypes = [C.POINTER(TVar), C.POINTER(TVar)]
x0 = W.c3(int(sys.argv[1]) if len(sys.argv) > 1 else 512, seed=512)
n = x0.size
def tv(a):
    v = TVar(); v.datatype = 1; v.buf.d = a.ctypes.data_as(C.POINTER(C.c_double)); return v
for rep in range(3):
    x = x0.copy(); z = np.zeros(n * 8 + 4096, np.uint8); rec = np.zeros(n)
    var, vz, vr = tv(x), TVar(), tv(rec)
    vz.datatype = 1; vz.buf.d = z.ctypes.data_as(C.POINTER(C.c_double))
    out = C.c_size_t(0)
    t0 = time.perf_counter(); lib.dctz_compress(C.byref(var), n, C.byref(out), C.byref(vz), 1e-3); t1 = time.perf_counter()
    lib.dctz_decompress(C.byref(vz), C.byref(vr)); t2 = time.perf_counter()
    print(f"rep {rep}: compress {1e3 * (t1 - t0):.1f} ms, decompress {1e3 * (t2 - t1):.1f} ms", flush=True)
