#!/usr/bin/env python3
"""End-to-end timing of the DROP-IN calls dctz_compress() / dctz_decompress() (host buffers in,
.z container out): H2D + GPU stage + D2H + the host zlib tail, with the library's own stage
timers (dctz_last_stage_times = the reference's TIME_DEBUG split, dctz-comp-lib.c:762-773).

SURVEY.md section 8(d) "timing protocol": reported beside the headline, never the headline.
Runs the reference's tail (three single-shot deflates), the chunked tail (DCTZ_ZLIB_THREADS = host cores)
and the entropy stage on the device (DCTZ_ZLIB_GPU=1) on the same input and checks that all containers
inflate to the same streams.  Prints one JSON object.

  python tools/e2e_bench.py [--n 512] [--dtype f64] [--eb 1e-3] [--mode ec] [--threads 16]
"""
import argparse
import ctypes as C
import json
import os
import struct
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class TVarBuf(C.Union):
    _fields_ = [("f", C.POINTER(C.c_float)), ("d", C.POINTER(C.c_double))]


class TVar(C.Structure):   # dctz.h:49-59
    _fields_ = [("datatype", C.c_int), ("err_bound", C.c_double), ("var_name", C.c_char_p), ("buf", TVarBuf)]


class StageTimes(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("h2d_s", "gpu_s", "d2h_s", "zlib_s", "total_s")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--eb", type=float, default=1e-3)
    ap.add_argument("--mode", choices=["ec", "qt"], default="ec")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--skip-reference-tail", action="store_true", help="leave out the three single-shot deflates (7 s per GiB)")
    a = ap.parse_args()
    import numpy as np
    from tests import workloads as W

    os.environ["DCTZ_QUIET"] = "1"
    lib = C.CDLL(os.path.join(ROOT, "dctz_amd", "lib", f"libdctz-{a.mode}.so"))
    lib.dctz_compress.argtypes = [C.POINTER(TVar), C.c_int, C.POINTER(C.c_size_t), C.POINTER(TVar), C.c_double]
    lib.dctz_decompress.argtypes = [C.POINTER(TVar), C.POINTER(TVar)]
    lib.dctz_last_stage_times.argtypes = [C.POINTER(StageTimes)]

    dt = np.float64 if a.dtype == "f64" else np.float32
    x0 = W.c3(a.n, seed=512, dtype=dt)
    n = x0.size

    def tvar(arr):
        v = TVar()
        v.datatype = 1 if arr.dtype == np.float64 else 0
        if arr.dtype == np.float64:
            v.buf.d = arr.ctypes.data_as(C.POINTER(C.c_double))
        else:
            v.buf.f = arr.ctypes.data_as(C.POINTER(C.c_float))
        return v

    def one(threads):
        os.environ.pop("DCTZ_ZLIB_GPU", None)
        os.environ.pop("DCTZ_ZLIB_THREADS", None)
        os.environ.pop("DCTZ_FAST_MEAN", None)
        os.environ.pop("DCTZ_PIPELINE", None)
        if threads in ("gpu", "gpu_fast_mean", "gpu_fast_mean_serial"):
            os.environ["DCTZ_ZLIB_GPU"] = "1"
            if threads != "gpu":
                os.environ["DCTZ_FAST_MEAN"] = "1"
            if threads == "gpu_fast_mean_serial":              # round 3's calls: every stage after the other
                os.environ["DCTZ_PIPELINE"] = "0"
        elif threads:
            os.environ["DCTZ_ZLIB_THREADS"] = str(threads)
        res = {}
        streams = None
        ctimes, dtimes = [], []
        for rep in range(5):                      # first repetition warms the context / page tables; the box is shared: the best of
                                                  # the others is reported, all of them are listed
            x = x0.copy()
            zbuf = np.zeros(n * x.itemsize + 4096, np.uint8)
            rec = np.zeros(n, dt)
            var, var_z, var_r = tvar(x), TVar(), tvar(rec)
            var_z.datatype = var.datatype
            var_z.buf.d = zbuf.ctypes.data_as(C.POINTER(C.c_double))
            out = C.c_size_t(0)
            st = StageTimes()
            t0 = time.perf_counter()
            lib.dctz_compress(C.byref(var), n, C.byref(out), C.byref(var_z), a.eb)
            t1 = time.perf_counter()
            lib.dctz_last_stage_times(C.byref(st))
            comp = {k: getattr(st, k) for k, _ in StageTimes._fields_}
            t2 = time.perf_counter()
            lib.dctz_decompress(C.byref(var_z), C.byref(var_r))
            t3 = time.perf_counter()
            lib.dctz_last_stage_times(C.byref(st))
            dec = {k: getattr(st, k) for k, _ in StageTimes._fields_}
            if rep:
                ctimes.append(t1 - t0)
                dtimes.append(t3 - t2)
            if rep and (not res or t1 - t0 <= min(ctimes)):
                res = {"compress_s": t1 - t0, "decompress_s": min(dtimes), "out_bytes": out.value,
                       "compress_stages_s": comp, "decompress_stages_s": dec,
                       "max_abs_err_vs_scaled_input": float(np.abs(rec - x0).max())}
            if rep == 4:
                res["decompress_s"] = min(dtimes)
                res["compress_GBps_input"] = x.nbytes / res["compress_s"] / 1e9
                res["decompress_GBps_input"] = x.nbytes / res["decompress_s"] / 1e9
                res["all_compress_ms"] = [round(t * 1e3, 2) for t in ctimes]
                res["all_decompress_ms"] = [round(t * 1e3, 2) for t in dtimes]
            z = zbuf[:out.value].tobytes()
            s0, s1, s2 = struct.unpack_from("<III", z, 40)
            streams = (zlib.decompress(z[56:56 + s0]), zlib.decompress(z[56 + s0:56 + s0 + s1]),
                       zlib.decompress(z[56 + s0 + s1:56 + s0 + s1 + s2]))
        return res, streams

    ref, s_ref = (one(0) if not a.skip_reference_tail else (None, None))
    par, s_par = one(a.threads)
    gpu, s_gpu = one("gpu")
    gpu_fm, s_gpu_fm = one("gpu_fast_mean")
    gpu_ser, s_gpu_ser = one("gpu_fast_mean_serial")
    assert (s_ref is None or s_ref == s_par) and s_par == s_gpu == s_gpu_fm == s_gpu_ser, "all tails must inflate to the same three streams"
    print(json.dumps({"what": "drop-in dctz_compress/dctz_decompress, host buffers, zlib included",
                      "workload": f"{a.dtype} {a.n}^3 C3 formula, {a.mode.upper()} eb={a.eb:g}", "input_bytes": x0.nbytes,
                      "host_cores": os.cpu_count(), "zlib": zlib.ZLIB_VERSION,
                      "reference_tail_3_threads": ref, f"chunked_tail_{a.threads}_threads": par, "deflate_on_gpu": gpu, "deflate_on_gpu_tree_order_mean": gpu_fm,
                      "deflate_on_gpu_tree_order_mean_stages_not_overlapped": gpu_ser,
                      "streams_identical": True}))


if __name__ == "__main__":
    main()
