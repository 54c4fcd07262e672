"""Host-side zlib tail: the chunked multi-threaded deflate must produce ONE standard zlib
stream per section (the reference reader is inflateInit + inflate, dctz-decomp-lib.c:244-322),
for every size / chunk / thread combination, including the empty section (cnt == 0)."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "dctz_amd", "lib", "libdctz-ec.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    L = C.CDLL(so)
    L.dctz_pdeflate_bound.restype = C.c_size_t
    L.dctz_pdeflate_bound.argtypes = [C.c_size_t, C.c_size_t]
    L.dctz_pdeflate.restype = C.c_int
    L.dctz_pdeflate.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int, C.c_size_t]
    return L


def _pdeflate(L, data, threads, chunk):
    cap = L.dctz_pdeflate_bound(len(data), chunk)
    dst = (C.c_ubyte * cap)()
    out_len = C.c_size_t(0)
    src = (C.c_ubyte * max(1, len(data))).from_buffer_copy(data if len(data) else b"\0")
    rc = L.dctz_pdeflate(src, len(data), dst, cap, C.byref(out_len), threads, chunk)
    assert rc == 0
    assert out_len.value <= cap
    return bytes(dst[:out_len.value])


def _bin_like(n, seed):
    """bin_index-like bytes: mostly small values, some 255 markers."""
    rng = np.random.default_rng(seed)
    b = np.abs(rng.normal(0, 3, n)).astype(np.uint8)
    b[rng.random(n) < 0.05] = 255
    return b.tobytes()


@pytest.mark.parametrize("n", [0, 1, 5, 32767, 32768, 32769, 65536, 100000, 1 << 20, (1 << 20) + 3])
@pytest.mark.parametrize("threads,chunk", [(1, 32768), (4, 32768), (8, 65536), (3, 1 << 18)])
def test_one_standard_zlib_stream(lib, n, threads, chunk):
    data = _bin_like(n, n + threads)
    z = _pdeflate(lib, data, threads, chunk)
    assert z[:2] == b"\x78\x9c"
    d = zlib.decompressobj()
    out = d.decompress(z)
    assert d.eof and d.unused_data == b"", "exactly one complete zlib stream, adler32 verified by inflate"
    assert out == data


def test_ratio_close_to_single_shot(lib):
    data = _bin_like(4 << 20, 3)
    single = len(zlib.compress(data, 6))
    chunked = len(_pdeflate(lib, data, 8, 1 << 18))
    assert chunked <= single * 1.02, (single, chunked)


def test_deterministic_across_thread_counts(lib):
    data = _bin_like(3 << 20, 5)
    a = _pdeflate(lib, data, 1, 1 << 18)
    b = _pdeflate(lib, data, 7, 1 << 18)
    assert a == b, "output depends on the chunking only, never on scheduling"


def test_level_knob(lib):
    lib.dctz_pdeflate_set_level.argtypes = [C.c_int]
    data = _bin_like(2 << 20, 9)
    lib.dctz_pdeflate_set_level(1)
    fast = _pdeflate(lib, data, 4, 1 << 18)
    lib.dctz_pdeflate_set_level(-1)
    dflt = _pdeflate(lib, data, 4, 1 << 18)
    assert zlib.decompress(fast) == data and zlib.decompress(dflt) == data
    assert len(fast) >= len(dflt)
