"""ctypes binding of lib/libdctzhip.so (include/dctz_hip.h).

torch is used only for device buffers and the stream; every compute call goes
through the C ABI.  No fallback: a missing library or GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
F32, F64 = 0, 1
EC, QT = 0, 1
OK, E_ARG, E_BOUND, E_HIP, E_INTERNAL = 0, -1, -2, -3, -4      # DCTZHIP_OK / DCTZHIP_E_*


class DctzHipError(RuntimeError):
    pass


def lib_path():
    return os.environ.get("DCTZHIP_LIBRARY") or os.path.join(_HERE, "lib", "libdctzhip.so")   # override: A/B builds of the library


class CompressInfo(C.Structure):
    _fields_ = [("sf", C.c_double), ("mean", C.c_double), ("max_abs", C.c_double),
                ("min_abs", C.c_double), ("cnt", C.c_uint32), ("nblk", C.c_uint32),
                ("qtable", C.c_double * 64), ("qtable_raw", C.c_double * 64),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


INFO_STATS_FUSED = 1   # sampled guess of sf verified: the separate statistics pass was saved
INFO_RESPUN = 2        # guess wrong: the compress kernels ran a second time with the true statistics
INFO_SPLIT = 8         # the compress kernel was k_compress_eo (a block over two lanes)
INFO_SINGLE_PASS = 16  # ... and AC_exact was placed by that kernel itself (no k_compact_ac)
INFO_LB_FALLBACK = 32  # ... whose look-back gave up: the pass ran again through the lists
INFO_ONE_LAUNCH = 4    # the whole call was one kernel (arrays whose tiles are all resident at once: dctz_kernels_one.hip)


class Timings(C.Structure):
    _fields_ = [("stats_ms", C.c_float), ("main_ms", C.c_float), ("tail_ms", C.c_float),
                ("total_ms", C.c_float)]


class BatchCItem(C.Structure):       # dctzhip_batch_citem
    _fields_ = [("d_in", C.c_void_p), ("n", C.c_size_t), ("dtype", C.c_int), ("error_bound", C.c_double),
                ("d_bin_index", C.c_void_p), ("d_dc", C.c_void_p), ("d_ac_exact", C.c_void_p), ("d_scaled", C.c_void_p)]


class BatchDItem(C.Structure):       # dctzhip_batch_ditem
    _fields_ = [("d_bin_index", C.c_void_p), ("d_dc", C.c_void_p), ("d_ac_exact", C.c_void_p), ("ac_count", C.c_uint32),
                ("qtable_host", C.c_void_p), ("n", C.c_size_t), ("dtype", C.c_int), ("error_bound", C.c_double),
                ("sf", C.c_double), ("d_out", C.c_void_p)]


_lib = None

_PROTOS = {
    "dctzhip_device_count": (C.c_int, []),
    "dctzhip_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "dctzhip_ctx_destroy": (None, [C.c_void_p]),
    "dctzhip_last_error": (C.c_char_p, [C.c_void_p]),
    "dctzhip_reserve": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "dctzhip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dctzhip_use_own_stream": (C.c_int, [C.c_void_p]),
    "dctzhip_get_stream": (C.c_void_p, [C.c_void_p]),
    "dctzhip_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "dctzhip_last_timings": (C.c_int, [C.c_void_p, C.POINTER(Timings)]),
    "dctzhip_set_speculation": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "dctzhip_set_one_launch": (C.c_int, [C.c_void_p, C.c_int]),
    "dctzhip_set_split": (C.c_int, [C.c_void_p, C.c_int]),
    "dctzhip_debug_counter": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]),
    "dctzhip_debug_knob": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "dctzhip_debug_last_kernel": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]),
    "dctzhip_malloc": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "dctzhip_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dctzhip_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dctzhip_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dctzhip_sync": (C.c_int, [C.c_void_p]),
    "dctzhip_host_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dctzhip_host_unregister": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dctzhip_compress": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(CompressInfo)]),
    "dctzhip_decompress": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                     C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_double, C.c_int,
                                     C.c_void_p]),
    "dctzhip_set_blocking": (C.c_int, [C.c_void_p, C.c_int]),
    "dctzhip_compress_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(BatchCItem), C.c_int, C.POINTER(CompressInfo)]),
    "dctzhip_decompress_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(BatchDItem), C.c_int, C.POINTER(C.c_int)]),
    "dctzhip_last_batch_timings": (C.c_int, [C.c_void_p, C.POINTER(Timings)]),
    "dctzhip_nd_blocks": (C.c_size_t, [C.c_int, C.POINTER(C.c_size_t)]),
    "dctzhip_compress_nd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.c_int, C.c_double, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(CompressInfo)]),
    "dctzhip_decompress_nd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_size_t), C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "dctzhip_dct_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "dctzhip_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(CompressInfo)]),
    "dctzhip_serial_mean_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "dctzhip_serial_mean_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "dctzhip_scale_inplace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double]),
    "dctzhip_debug_divide": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_void_p,
                                       C.c_void_p]),
    "dctzhip_psnr_terms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "dctzhip_deflate_bound": (C.c_size_t, [C.c_size_t]),
    "dctzhip_deflate_chunk_bytes": (C.c_size_t, []),
    "dctzhip_deflate_ex": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_uint)]),
    "dctzhip_inflate": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "dctzhip_deflate": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]),
    "dctzhip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dctzhip_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dctzhip_comm_destroy": (C.c_int, [C.c_void_p]),
    "dctzhip_comm_sizes": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "dctzhip_comm_gather": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "dctzhip_compress_part": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dctzhip_h2d_pipe_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "dctzhip_h2d_pipe_wait": (C.c_int, [C.c_void_p, C.c_size_t]),
    "dctzhip_h2d_pipe_landed": (C.c_int, [C.c_void_p, C.c_size_t]),
    "dctzhip_h2d_pipe_end": (C.c_int, [C.c_void_p, C.c_int]),
    "dctzhip_version": (C.c_char_p, []),
}


def load_library():
    """Loads libdctzhip.so; raises DctzHipError if it has not been built."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise DctzHipError(f"{p} is missing: run `make -C dctz_amd` (or __graft_entry__.build()); "
                               "there is no CPU fallback")
        lib = C.CDLL(p)
        for name, (res, args) in _PROTOS.items():
            try:
                fn = getattr(lib, name)      # AttributeError if the ABI lost a symbol
            except AttributeError:
                if os.environ.get("DCTZHIP_LIBRARY"):      # an older A/B build of the library: that entry point is just absent
                    continue
                raise
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _dt(torch_dtype):
    import torch
    if torch_dtype == torch.float64:
        return F64
    if torch_dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {torch_dtype}")


class Context:
    """One dctzhip context on a GPU; kernels run on torch's current stream (bound before every call: the raw
    handle, where 0 = the legacy default stream, goes to dctzhip_set_stream as is)."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise DctzHipError("no GPU visible: the DCTZ hot path has no CPU fallback")
        self.lib = load_library()
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        rc = self.lib.dctzhip_ctx_create(C.byref(h), device)
        if rc != 0:
            raise DctzHipError(f"dctzhip_ctx_create: {self.lib.dctzhip_last_error(None).decode()}")
        self.h = h
        self._bound = object()            # nothing bound yet
        self._bind_stream()

    def _bind_stream(self):
        # the raw handle of torch's current stream on this device (the private accessor skips the
        # Stream object; fall back to the public API if it is not there)
        try:
            s = self.torch._C._cuda_getCurrentRawStream(self.device.index)
        except AttributeError:
            s = self.torch.cuda.current_stream(self.device).cuda_stream
        if s != self._bound:
            self.lib.dctzhip_set_stream(self.h, C.c_void_p(s))
            self._bound = s

    def close(self):
        if getattr(self, "h", None):
            self.lib.dctzhip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise DctzHipError(f"{what} failed ({rc}): {self.lib.dctzhip_last_error(self.h).decode()}")

    def set_profiling(self, on=True):
        self._check(self.lib.dctzhip_set_profiling(self.h, int(on)), "set_profiling")

    def set_speculation(self, on=True, min_elements=0):
        """Fused statistics behind a sampled guess of sf (include/dctz_hip.h, DCTZHIP_INFO_*)."""
        self._check(self.lib.dctzhip_set_speculation(self.h, int(on), int(min_elements)), "set_speculation")

    def set_one_launch(self, on=True):
        """One kernel per call for arrays whose tiles are all resident at once (include/dctz_hip.h); off: the chain of kernels."""
        self._check(self.lib.dctzhip_set_one_launch(self.h, int(on)), "set_one_launch")

    def set_split(self, on=True):
        """k_compress_eo (a block over two lanes) for flat fp64 arrays on the chain of kernels (include/dctz_hip.h)."""
        self._check(self.lib.dctzhip_set_split(self.h, int(on)), "set_split")

    def counter(self, which):
        """dctzhip_debug_counter (include/dctz_hip.h): 0 one-launch calls, 1 launches that gave up, 2 cooldown, 3 split calls, ..."""
        v = C.c_ulonglong(0)
        self._check(self.lib.dctzhip_debug_counter(self.h, int(which), C.byref(v)), "debug_counter")
        return int(v.value)

    def last_kernel(self, which):
        """The name rocprofv3 lists the big kernel of the last call under (0 compress, 1 decompress, 2-5 batches)."""
        buf = C.create_string_buffer(128)
        self._check(self.lib.dctzhip_debug_last_kernel(self.h, int(which), buf, 128), "debug_last_kernel")
        return buf.value.decode()

    def knob(self, key, value):
        self._check(self.lib.dctzhip_debug_knob(self.h, int(key), int(value)), "debug_knob")

    def set_blocking(self, on=True):
        """Calls return only when their outputs are complete for any observer (default: complete in stream order)."""
        self._check(self.lib.dctzhip_set_blocking(self.h, int(on)), "set_blocking")

    def timings(self):
        t = Timings()
        self._check(self.lib.dctzhip_last_timings(self.h, C.byref(t)), "last_timings")
        return {"stats_ms": t.stats_ms, "main_ms": t.main_ms, "tail_ms": t.tail_ms, "total_ms": t.total_ms}

    def reserve(self, n, dtype, mode):
        self._check(self.lib.dctzhip_reserve(self.h, n, _dt(dtype), mode), "reserve")

    def alloc_outputs(self, n, dtype=None):
        t = self.torch
        nblk = (n + 63) // 64
        return {"bin_index": t.empty(n, dtype=t.uint8, device=self.device),
                "dc": t.empty(nblk, dtype=t.float32, device=self.device),
                "ac_exact": t.empty(n, dtype=t.float32, device=self.device)}

    def compress(self, x, eb, mode=EC, out=None, scaled=None, coef=None):
        """x: 1-D contiguous CUDA tensor (float32|float64).  Returns (out, info)."""
        assert x.is_cuda and x.is_contiguous() and x.dim() == 1
        self._bind_stream()
        n = x.numel()
        if out is None:
            out = self.alloc_outputs(n, x.dtype)
        info = CompressInfo()
        rc = self.lib.dctzhip_compress(
            self.h, x.data_ptr(), n, _dt(x.dtype), float(eb), mode, out["bin_index"].data_ptr(),
            out["dc"].data_ptr(), out["ac_exact"].data_ptr(),
            scaled.data_ptr() if scaled is not None else None,
            coef.data_ptr() if coef is not None else None, C.byref(info))
        self._check(rc, "dctzhip_compress")
        return out, info

    # ---- the same two calls on fixed buffers, their arguments converted once (timing loops: what a C caller's loop costs,
    # without a dozen tensor-attribute look-ups and a fresh 1 KiB info structure per call on the Python side)
    def prepare_pair(self, x, out, dst, eb, mode=EC):
        assert x.is_cuda and x.is_contiguous() and x.dim() == 1 and dst.numel() == x.numel() and dst.dtype == x.dtype
        self._bind_stream()
        info = CompressInfo()
        vp, n, dt = C.c_void_p, x.numel(), _dt(x.dtype)
        b, d, a = vp(out["bin_index"].data_ptr()), vp(out["dc"].data_ptr()), vp(out["ac_exact"].data_ptr())
        cargs = (self.h, vp(x.data_ptr()), C.c_size_t(n), C.c_int(dt), C.c_double(float(eb)), C.c_int(mode), b, d, a, None, None, C.byref(info))
        qtab = None
        if mode == QT:                                   # the call's table in the element type, refreshed from info per call
            qtab = (C.c_double * 64)() if x.dtype == self.torch.float64 else (C.c_float * 64)()
        dhead = (self.h, b, d, a)
        dtail = (C.c_size_t(n), C.c_int(dt), C.c_double(float(eb)))
        return {"info": info, "cargs": cargs, "dhead": dhead, "dtail": dtail, "qtab": qtab, "mode": C.c_int(mode), "dst": vp(dst.data_ptr()),
                "keep": (x, out, dst)}

    def compress_prepared(self, pp):
        rc = self.lib.dctzhip_compress(*pp["cargs"])
        if rc:
            self._check(rc, "dctzhip_compress")
        return pp["info"]

    def decompress_prepared(self, pp):
        info, q = pp["info"], pp["qtab"]
        if q is not None:
            q[:] = info.qtable[:]
        rc = self.lib.dctzhip_decompress(*pp["dhead"], info.cnt, q, *pp["dtail"], info.sf, pp["mode"], pp["dst"])
        if rc:
            self._check(rc, "dctzhip_decompress")

    def compress_part(self, x, eb, max_abs, min_abs, out, lo, ac_at):
        """Elements of one part of an array (x: the part, a 1-D CUDA tensor that starts on a block boundary `lo` of the
        array) with the ARRAY's max|x| / min|x|: its streams go to their places in `out` (the array's outputs), the exact
        coefficients at ac_at.  Returns (cnt, (max|x|, min|x|, sum from the second element on), sf)."""
        assert x.is_cuda and x.is_contiguous() and x.dim() == 1 and lo % 64 == 0
        self._bind_stream()
        cnt, st, sf = C.c_uint32(0), (C.c_double * 3)(), C.c_double(0.0)
        rc = self.lib.dctzhip_compress_part(
            self.h, x.data_ptr(), x.numel(), _dt(x.dtype), float(eb), float(max_abs), float(min_abs),
            out["bin_index"].data_ptr() + lo, out["dc"].data_ptr() + 4 * (lo // 64), out["ac_exact"].data_ptr() + 4 * int(ac_at),
            C.byref(cnt), st, C.byref(sf))
        self._check(rc, "dctzhip_compress_part")
        return cnt.value, (st[0], st[1], st[2]), sf.value

    def decompress(self, out, cnt, n, dtype, eb, sf, mode=EC, qtable=None, dst=None):
        t = self.torch
        self._bind_stream()
        if dst is None:
            dst = t.empty(n, dtype=dtype, device=self.device)
        q = None
        if mode == QT:
            q = np.ascontiguousarray(qtable, dtype=np.float64 if dtype == t.float64 else np.float32)
            assert q.size == 64
        rc = self.lib.dctzhip_decompress(
            self.h, out["bin_index"].data_ptr(), out["dc"].data_ptr(), out["ac_exact"].data_ptr(),
            int(cnt), q.ctypes.data_as(C.c_void_p) if q is not None else None, n, _dt(dtype),
            float(eb), float(sf), mode, dst.data_ptr())
        self._check(rc, "dctzhip_decompress")
        return dst

    # ---- batches of arrays (include/dctz_hip.h: dctzhip_compress_batch / dctzhip_decompress_batch) ----
    def compress_batch(self, xs, ebs, mode=EC, outs=None, scaled=None, prepared=None):
        """k arrays (1-D contiguous CUDA tensors, float32 | float64 each) through one launch sequence per element type.
        ebs: one bound or one per array.  Returns (outs, infos, prepared); pass `prepared` back in to repeat the very
        same call without rebuilding the argument table."""
        self._bind_stream()
        if prepared is None:
            k = len(xs)
            ebs = [float(ebs)] * k if not hasattr(ebs, "__len__") else [float(e) for e in ebs]
            if outs is None:
                outs = [self.alloc_outputs(x.numel(), x.dtype) for x in xs]
            items = (BatchCItem * max(k, 1))()
            for i, x in enumerate(xs):
                assert x.is_cuda and x.is_contiguous() and x.dim() == 1
                it = items[i]
                it.d_in, it.n, it.dtype, it.error_bound = x.data_ptr(), x.numel(), _dt(x.dtype), ebs[i]
                it.d_bin_index, it.d_dc, it.d_ac_exact = outs[i]["bin_index"].data_ptr(), outs[i]["dc"].data_ptr(), outs[i]["ac_exact"].data_ptr()
                it.d_scaled = scaled[i].data_ptr() if (scaled is not None and scaled[i] is not None) else None
            prepared = (k, items, (CompressInfo * max(k, 1))(), outs, (xs, scaled))       # (keeps the tensors alive)
        k, items, infos, outs, _ = prepared
        self._check(self.lib.dctzhip_compress_batch(self.h, k, items, mode, infos), "dctzhip_compress_batch")
        return outs, [infos[i] for i in range(k)], prepared

    def decompress_batch(self, outs, cnts, ns, dtypes, ebs, sfs, mode=EC, qtables=None, dsts=None, prepared=None, check=True):
        """The decode side of a batch.  Returns (dsts, status, prepared); raises on an under-run unless check=False."""
        t = self.torch
        self._bind_stream()
        if prepared is None:
            k = len(outs)
            ebs = [float(ebs)] * k if not hasattr(ebs, "__len__") else [float(e) for e in ebs]
            if dsts is None:
                dsts = [t.empty(int(ns[i]), dtype=dtypes[i], device=self.device) for i in range(k)]
            items = (BatchDItem * max(k, 1))()
            keep = []
            for i in range(k):
                it = items[i]
                it.d_bin_index, it.d_dc, it.d_ac_exact = outs[i]["bin_index"].data_ptr(), outs[i]["dc"].data_ptr(), outs[i]["ac_exact"].data_ptr()
                it.ac_count, it.n, it.dtype, it.error_bound, it.sf = int(cnts[i]), int(ns[i]), _dt(dtypes[i]), ebs[i], float(sfs[i])
                it.d_out = dsts[i].data_ptr()
                if mode == QT:
                    q = np.ascontiguousarray(qtables[i], dtype=np.float64 if dtypes[i] == t.float64 else np.float32)
                    assert q.size == 64
                    keep.append(q)
                    it.qtable_host = q.ctypes.data
            prepared = (k, items, (C.c_int * max(k, 1))(), dsts, (outs, keep))
        k, items, status, dsts, _ = prepared
        rc = self.lib.dctzhip_decompress_batch(self.h, k, items, mode, status)
        if check:
            self._check(rc, "dctzhip_decompress_batch")
        return dsts, [status[i] for i in range(k)], prepared

    def batch_timings(self):
        """Device time of the last batch call per element-type sequence: {"f32": {...}, "f64": {...}} (profiling on)."""
        tt = (Timings * 2)()
        self._check(self.lib.dctzhip_last_batch_timings(self.h, tt), "last_batch_timings")
        return {name: {"stats_ms": tt[i].stats_ms, "main_ms": tt[i].main_ms, "tail_ms": tt[i].tail_ms, "total_ms": tt[i].total_ms}
                for i, name in ((F32, "f32"), (F64, "f64"))}

    # ---- multi-dimensional blocks (include/dctz_hip.h: 8 x 8 tiles of a 2-D array, 4 x 4 x 4 tiles of a 3-D array) ----
    def nd_blocks(self, shape):
        dims = (C.c_size_t * len(shape))(*shape)
        nblk = self.lib.dctzhip_nd_blocks(len(shape), dims)
        if nblk == 0:
            raise DctzHipError(f"multi-dimensional blocks: bad shape {tuple(shape)}")
        return nblk

    def compress_nd(self, x, eb, mode=EC, out=None, scaled=None):
        """x: contiguous 2-D or 3-D CUDA tensor.  The streams cover nblk * 64 positions (edge tiles are padded)."""
        assert x.is_cuda and x.is_contiguous() and x.dim() in (2, 3)
        self._bind_stream()
        shape = tuple(x.shape)
        if out is None:
            out = self.alloc_outputs(self.nd_blocks(shape) * 64, x.dtype)
        dims = (C.c_size_t * len(shape))(*shape)
        info = CompressInfo()
        rc = self.lib.dctzhip_compress_nd(
            self.h, x.data_ptr(), len(shape), dims, _dt(x.dtype), float(eb), mode, out["bin_index"].data_ptr(),
            out["dc"].data_ptr(), out["ac_exact"].data_ptr(), scaled.data_ptr() if scaled is not None else None,
            C.byref(info))
        self._check(rc, "dctzhip_compress_nd")
        return out, info

    def decompress_nd(self, out, cnt, shape, dtype, eb, sf, mode=EC, qtable=None, dst=None):
        t = self.torch
        self._bind_stream()
        shape = tuple(shape)
        if dst is None:
            dst = t.empty(shape, dtype=dtype, device=self.device)
        q = None
        if mode == QT:
            q = np.ascontiguousarray(qtable, dtype=np.float64 if dtype == t.float64 else np.float32)
            assert q.size == 64
        dims = (C.c_size_t * len(shape))(*shape)
        rc = self.lib.dctzhip_decompress_nd(
            self.h, out["bin_index"].data_ptr(), out["dc"].data_ptr(), out["ac_exact"].data_ptr(), int(cnt),
            q.ctypes.data_as(C.c_void_p) if q is not None else None, len(shape), dims, _dt(dtype), float(eb), float(sf),
            mode, dst.data_ptr())
        self._check(rc, "dctzhip_decompress_nd")
        return dst

    def psnr_terms(self, x, r):
        """calc_psnr's reductions (util.c:54-104) on the GPU: (min x, max x, max |x - r|, sum (x - r)^2)."""
        assert x.is_cuda and r.is_cuda and x.dtype == r.dtype and x.numel() == r.numel()
        self._bind_stream()
        out = (C.c_double * 4)()
        rc = self.lib.dctzhip_psnr_terms(self.h, x.data_ptr(), r.data_ptr(), x.numel(), _dt(x.dtype), out)
        self._check(rc, "dctzhip_psnr_terms")
        return tuple(out)

    def deflate(self, sections, want_index=False, literals=None):
        """zlib streams of byte sections, made on the GPU (dctzhip_deflate).  sections: device tensors (any dtype,
        contiguous); returns a list of uint8 device tensors holding one zlib stream each -- with want_index also the
        list of per-chunk compressed sizes (numpy uint32) of every section.  literals: per section, True = no match search
        (DCTZHIP_DEFLATE_LITERALS: bytes of floats)."""
        import torch
        self._bind_stream()
        k = len(sections)
        nbytes = [t.numel() * t.element_size() for t in sections]
        outs = [torch.empty(int(self.lib.dctzhip_deflate_bound(nb)), dtype=torch.uint8, device=sections[0].device if k else "cuda")
                for nb in nbytes]
        src = (C.c_void_p * max(k, 1))(*[t.data_ptr() if nb else None for t, nb in zip(sections, nbytes)])
        dst = (C.c_void_p * max(k, 1))(*[o.data_ptr() for o in outs])
        n = (C.c_size_t * max(k, 1))(*nbytes)
        cap = (C.c_size_t * max(k, 1))(*[o.numel() for o in outs])
        ln = (C.c_size_t * max(k, 1))()
        chunk = int(self.lib.dctzhip_deflate_chunk_bytes())
        idx = [np.zeros(max(1, (nb + chunk - 1) // chunk), np.uint32) for nb in nbytes]
        ix = (C.c_void_p * max(k, 1))(*[a.ctypes.data for a in idx])
        fl = (C.c_uint * max(k, 1))(*[1 if (literals and literals[i]) else 0 for i in range(k)])
        rc = self.lib.dctzhip_deflate_ex(self.h, k, src, n, dst, cap, ln, ix if want_index else None, fl)
        self._check(rc, "dctzhip_deflate")
        zs = [o[:int(l)] for o, l in zip(outs, ln)]
        if want_index:
            return zs, [a[:(nb + chunk - 1) // chunk] for a, nb in zip(idx, nbytes)]
        return zs

    def inflate(self, streams, index, raw_bytes):
        """Sections made by deflate() back to bytes on the GPU (dctzhip_inflate).  streams: uint8 device tensors; index:
        per-chunk compressed sizes (numpy uint32) per section; raw_bytes: inflated sizes.  Returns (list of uint8 device
        tensors, ok)."""
        import torch
        self._bind_stream()
        k = len(streams)
        outs = [torch.empty(max(int(nb), 1), dtype=torch.uint8, device=self.device) for nb in raw_bytes]
        idx = [np.ascontiguousarray(a, dtype=np.uint32) for a in index]
        z = (C.c_void_p * max(k, 1))(*[t.data_ptr() for t in streams])
        zl = (C.c_size_t * max(k, 1))(*[t.numel() for t in streams])
        ix = (C.c_void_p * max(k, 1))(*[a.ctypes.data if a.size else None for a in idx])
        raw = (C.c_size_t * max(k, 1))(*[int(nb) for nb in raw_bytes])
        dst = (C.c_void_p * max(k, 1))(*[o.data_ptr() for o in outs])
        ok = C.c_int(0)
        rc = self.lib.dctzhip_inflate(self.h, k, z, zl, ix, raw, dst, C.byref(ok))
        self._check(rc, "dctzhip_inflate")
        return [o[:int(nb)] for o, nb in zip(outs, raw_bytes)], bool(ok.value)

    # ---- multi-GPU gather of the pre-zlib streams over RCCL (include/dctz_hip.h, dctzhip_comm_*) ----
    @staticmethod
    def comm_unique_id():
        """Rank 0 makes the 128-byte id; the caller hands it to the other ranks."""
        lib = load_library()
        buf = C.create_string_buffer(128)
        if lib.dctzhip_comm_unique_id(buf) != 0:
            raise DctzHipError(f"dctzhip_comm_unique_id: {lib.dctzhip_last_error(None).decode()}")
        return buf.raw

    def comm_create(self, rank, world, unique_id):
        assert len(unique_id) == 128
        self._bind_stream()
        self._check(self.lib.dctzhip_comm_create(self.h, int(rank), int(world), C.c_char_p(unique_id)), "dctzhip_comm_create")
        self.comm = (int(rank), int(world))

    def comm_gather(self, out, cnt, n, root=0):
        """Collective.  Returns on `root` a dict of the concatenated streams of all ranks (device tensors) plus
        "sizes" = [(n, nblk, cnt)] per rank; None elsewhere."""
        t = self.torch
        rank, world = self.comm
        self._bind_stream()
        sizes = (C.c_uint64 * (3 * world))()
        self._check(self.lib.dctzhip_comm_sizes(self.h, int(n), int(cnt), sizes), "dctzhip_comm_sizes")
        sz = [(sizes[3 * r], sizes[3 * r + 1], sizes[3 * r + 2]) for r in range(world)]
        res = None
        ptrs = (None, None, None)
        if rank == root:
            # the receive buffers are kept between calls (the same shards every step: 1.3 GB of torch.empty per gather
            # otherwise) and only grow; the views handed back are valid until the next gather of this context
            need = (sum(s[0] for s in sz), sum(s[1] for s in sz), max(1, sum(s[2] for s in sz)))
            keep = getattr(self, "_gather_bufs", None)
            if keep is None or any(k.numel() < n_ for k, n_ in zip(keep, need)):
                keep = (t.empty(need[0], dtype=t.uint8, device=self.device), t.empty(need[1], dtype=t.float32, device=self.device),
                        t.empty(need[2], dtype=t.float32, device=self.device))
                self._gather_bufs = keep
            res = {"bin_index": keep[0][:need[0]], "dc": keep[1][:need[1]], "ac_exact": keep[2][:need[2]], "sizes": sz}
            ptrs = (keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr())
        self._check(self.lib.dctzhip_comm_gather(self.h, int(root), out["bin_index"].data_ptr(), out["dc"].data_ptr(),
                                                 out["ac_exact"].data_ptr(), sizes, *ptrs), "dctzhip_comm_gather")
        return res

    def debug_divide(self, x, divisor):
        self._bind_stream()
        fast, ref = self.torch.empty_like(x), self.torch.empty_like(x)
        rc = self.lib.dctzhip_debug_divide(self.h, x.data_ptr(), x.numel(), _dt(x.dtype), float(divisor),
                                           fast.data_ptr(), ref.data_ptr())
        self._check(rc, "dctzhip_debug_divide")
        return fast, ref

    def dct_blocks(self, x, inverse=False):
        self._bind_stream()
        y = self.torch.empty_like(x)
        rc = self.lib.dctzhip_dct_blocks(self.h, x.data_ptr(), y.data_ptr(), x.numel(), _dt(x.dtype), int(inverse))
        self._check(rc, "dctzhip_dct_blocks")
        return y
