"""Runs the PRODUCT's per-block transform (dctz_amd/csrc/dct64_block.h + dctz_tables.h: the code one GPU lane
executes, fused multiply-adds included) on the CPU (tests/emu/emu_dct64.cpp) and requires bit-identity with the
oracle's pinned fast flow.  This is what lets kernel-vs-oracle GPU comparisons be exact rather than tolerance-based;
independence from the kernel's own arithmetic comes from the oracle's definition-order flow and scipy (test_oracle.py,
test_noise_floor.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "emu", "emu_dct64.so")


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "emu", "emu_dct64.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", "-o", SO, src])
    return C.CDLL(SO)


@pytest.mark.parametrize("dtype,suf", [(np.float64, "f64"), (np.float32, "f32")])
def test_lane_flow_bit_identical_to_oracle(emu, dtype, suf):
    rng = np.random.default_rng(11)
    for i in range(1500):
        a = (rng.standard_normal(64) * 10 ** rng.uniform(-3, 2)).astype(dtype)
        if i == 0:
            a[:] = 0
        if i == 1:
            a[:] = 1
        b = np.empty_like(a)
        getattr(emu, "emu_fwd_" + suf)(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
        assert np.array_equal(b.view(np.uint8), O.dct_fwd(a, O.FAST).view(np.uint8))
        getattr(emu, "emu_inv_" + suf)(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
        assert np.array_equal(b.view(np.uint8), O.dct_inv(a, O.FAST).view(np.uint8))


@pytest.mark.parametrize("dtype,suf", [(np.float64, "f64"), (np.float32, "f32")])
def test_even_odd_halves_are_the_whole_transform(emu, dtype, suf):
    """dct64_block_eo.h: the half of dct64_fwd() behind the even-numbered coefficients and the half behind the odd-numbered
    ones (k_compress_eo gives them to two wavefronts) share no operation and together ARE dct64_fwd(): bit for bit the
    lane flow, hence the oracle's pinned flow.  Signed zeros, denormals and huge values included."""
    rng = np.random.default_rng(31)
    for i in range(3000):
        a = (rng.standard_normal(64) * 10 ** rng.uniform(-3, 2)).astype(dtype)
        if i == 0:
            a[:] = 0
        if i == 1:
            a[:] = -0.0
        if i == 2:
            a[:] = 1
        if i == 3:
            a *= dtype(np.finfo(dtype).tiny) * dtype(4)
        if i == 4:
            a *= dtype(np.finfo(dtype).max) / dtype(1e4)
        if i == 5:
            a[::2] = 0
        b, w = np.empty_like(a), np.empty_like(a)
        getattr(emu, "emu_eo_" + suf)(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
        getattr(emu, "emu_fwd_" + suf)(a.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p))
        assert np.array_equal(b.view(np.uint8), w.view(np.uint8)), i
        assert np.array_equal(b.view(np.uint8), O.dct_fwd(a, O.FAST).view(np.uint8)), i


@pytest.mark.parametrize("dtype,suf", [(np.float64, "f64"), (np.float32, "f32")])
def test_remainder_tables_identical_to_oracle(emu, dtype, suf):
    """Host tables of the product (explicit sincos) == the oracle's, for every length."""
    for l in range(1, 64):
        tab = np.zeros(512, dtype)
        getattr(emu, "emu_rem_tab_" + suf)(l, tab.ctypes.data_as(C.c_void_p))
        as_, ax, ias, iax = O.dct_tables(l, dtype)
        assert np.array_equal(tab[0:l], as_) and np.array_equal(tab[64:64 + l], ax)
        assert np.array_equal(tab[192:192 + l], iax) and np.array_equal(tab[129:128 + l], ias[1:])


@pytest.mark.parametrize("dtype,code,kmin,kmax", [(np.float64, 1, -324, 309), (np.float32, 0, -46, 39)])
def test_decade_tables_reproduce_the_host_scaling_factor(emu, dtype, code, kmin, kmax):
    """The device chooses the scaling factor of a speculative call as pw[#{k : thr[k] < max}] from tables the host
    builds with its own log10 / pow (dctz_tables.h: decade_tables).  That must be scaling_factor() -- util.c:29 / :43 --
    for every value: checked at both sides of every decade boundary, at exact powers of ten, and on random values; and
    against the oracle's calc_data_stat restatement."""
    nk = kmax - kmin + 1
    thr = np.zeros(nk)
    pw = np.zeros(nk + 1)
    emu.emu_decades.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    emu.emu_scaling_factor.restype = C.c_double
    emu.emu_scaling_factor.argtypes = [C.c_int, C.c_double]
    emu.emu_decades(code, kmin, kmax, thr.ctypes.data_as(C.c_void_p), pw.ctypes.data_as(C.c_void_p))
    assert np.all(np.diff(thr) >= 0)

    def dev(v):
        return pw[int((thr < v).sum())]

    probes = []
    for t in thr:
        if 0 < t < np.inf:
            probes += [t, float(np.nextafter(dtype(t), dtype(np.inf))), float(np.nextafter(dtype(t), dtype(0)))]
    probes += [float(dtype(10.0) ** k) for k in range(-20, 21)] + [1.0, 0.1, 0.5, 37.5, 999.9999, 1000.0, 1000.0001]
    rng = np.random.default_rng(9)
    probes += list((10.0 ** rng.uniform(-30, 30, 4000)).astype(dtype).astype(np.float64))
    fin = np.finfo(dtype)
    for v in probes:
        v = float(dtype(v))
        if not (0 < v <= fin.max):
            continue
        assert dev(v) == emu.emu_scaling_factor(code, v), v
    # ... and scaling_factor() itself is the oracle's (util.c:29 / :43 through the same libm)
    for v in (0.37, 1.0, 9.99, 10.0, 10.01, 37.5, 1e-7, 123456.0):
        assert emu.emu_scaling_factor(code, float(dtype(v))) == O.stats(np.array([0.0, v], dtype)).sf


def test_packed_fp32_transform_bit_identical_to_oracle(emu):
    """dct64_block_pk.h (two fp32 values per instruction on the GPU; GCC vectors here) performs, component for
    component, the operations of the scalar flow: forward and inverse == the oracle's pinned flow, bit for bit."""
    rng = np.random.default_rng(23)
    for i in range(2000):
        a = (rng.standard_normal(64) * 10 ** rng.uniform(-3, 2)).astype(np.float32)
        if i == 0:
            a[:] = 0
        if i == 1:
            a[:] = 1
        b = np.empty_like(a)
        emu.emu_pk_f32(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), 0)
        assert np.array_equal(b.view(np.uint32), O.dct_fwd(a, O.FAST).view(np.uint32))
        emu.emu_pk_f32(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), 1)
        assert np.array_equal(b.view(np.uint32), O.dct_inv(a, O.FAST).view(np.uint32))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_host_per_block_entry_points_follow_the_pinned_flow(dtype):
    """dct_fftw / ifft_idct of the drop-in run on the host (dctz_amd/csrc/dct_host.cpp: the product's lane flow compiled for
    the CPU).  The exported block routine against the oracle's pinned flow, every length 1 .. 64, both directions -- no GPU
    involved (the library loads without one; only its GPU entry points need one)."""
    import ctypes as C
    from oracle import oracle as O
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dctz_amd", "lib", "libdctz-ec.so")
    if not os.path.exists(so):
        pytest.skip("drop-in library not built")
    lib = C.CDLL(so)
    fn = lib.dctz_host_block_f64 if dtype == np.float64 else lib.dctz_host_block_f32
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    rng = np.random.default_rng(5)
    for l in range(1, 65):
        x = (rng.standard_normal(l) * 3.0).astype(dtype)
        a, b = np.zeros_like(x), np.zeros_like(x)
        fn(x.ctypes.data, a.ctypes.data, l, 0)
        assert np.array_equal(a.view(np.uint8), O.dct_fwd(x, O.FAST).view(np.uint8)), l
        fn(a.ctypes.data, b.ctypes.data, l, 1)
        assert np.array_equal(b.view(np.uint8), O.dct_inv(a, O.FAST).view(np.uint8)), l
        y = x.copy()
        fn(y.ctypes.data, y.ctypes.data, l, 0)                 # in place
        assert np.array_equal(y.view(np.uint8), a.view(np.uint8))
