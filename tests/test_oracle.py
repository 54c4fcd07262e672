"""Pins the CPU oracle: survey known answers (outputs of the reference itself,
SURVEY.md section 8c), scipy's independent orthonormal DCT, and structural identities."""
import json
import os
import zlib

import numpy as np
import pytest
import scipy.fft as sfft

from oracle import oracle as O
from tests import workloads as W

KA = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_known_answers.json")))


def _zsizes(c):
    # dctz-comp-lib.c:642-643: deflateInit2(level default(6), windowBits 15, memLevel 8)
    return [len(zlib.compress(a.tobytes(), 6)) for a in (c.bin_index, c.dc, c.ac_exact)]


@pytest.mark.parametrize("impl", [O.FAST, O.NAIVE])
def test_c1_ec_known_answer(impl):
    ka = KA["C1_ec"]
    x = W.c1()
    c = O.compress(x, ka["eb"], O.EC, impl)
    assert c.sf == ka["sf"]
    assert c.mean == ka["mean"]                       # bit-exact, incl. util.c:22 "i=1" quirk
    assert c.cnt == ka["cnt"]
    assert int((c.bin_index == 255).sum()) == ka["n255"]
    r = O.decompress(c, impl)
    p = O.psnr((x / c.sf) * c.sf, r)                  # dctz-test.c:188-210 un-scale
    assert abs(p["psnr"] - ka["psnr"]) / ka["psnr"] < 1e-11
    assert abs(p["maxdiff"] - ka["maxerr"]) / ka["maxerr"] < 1e-6
    assert abs(p["rmse"] - ka["rmse"]) / ka["rmse"] < 1e-9
    if impl == O.FAST and zlib.ZLIB_VERSION.startswith("1.2.11"):
        zs = _zsizes(c)
        assert zs == ka["zlib_sizes"]
        assert 56 + sum(zs) == ka["outSize"]


def test_c1_qt_known_answer():
    ka = KA["C1_qt"]
    x = W.c1()
    c = O.compress(x, ka["eb"], O.QT, O.FAST)
    assert c.cnt == ka["cnt"] and int((c.bin_index == 255).sum()) == ka["n255"]
    np.testing.assert_allclose(c.qtable[1:4], ka["q1_3"], rtol=5e-9)
    assert np.array_equal(c.qtable[1:], np.maximum(c.qtable_raw[1:], 1.0))
    assert c.qtable[0] == c.qtable_raw[0]             # slot 0 = last block's DC, unclamped
    r = O.decompress(c)
    p = O.psnr((x / c.sf) * c.sf, r)
    assert round(p["psnr"], 2) == ka["psnr_2dp"]


def test_c3_256_qt_known_answer():
    ka = KA["C3_256_qt"]
    v = W.c3(256)
    c = O.compress(v, ka["eb"], O.QT, O.FAST)
    assert c.sf == ka["sf"] and c.cnt == ka["cnt"]
    assert int((c.bin_index == 255).sum()) == ka["n255"]
    r = O.decompress(c)
    p = O.psnr((v / c.sf) * c.sf, r)
    assert round(p["psnr"], 2) == ka["psnr_2dp"]
    assert round(p["maxdiff"] / p["range"], 6) == ka["max_rel_err_6dp"]


@pytest.mark.parametrize("mode", ["ec", "qt"])
def test_c2_known_answer(mode):
    ka = KA["C2_" + mode]
    f = W.c2()
    c = O.compress(f, ka["eb"], O.QT if mode == "qt" else O.EC, O.FAST)
    assert c.sf == ka["sf"]
    r = O.decompress(c)
    p = O.psnr(f, r)
    assert round(p["psnr"], 2) == ka["psnr_2dp"]
    if zlib.ZLIB_VERSION.startswith("1.2.11"):
        tot = 56 + sum(_zsizes(c)) + (256 if mode == "qt" else 0)
        assert round(f.nbytes / tot, 2) == ka["cr_2dp"]


@pytest.mark.parametrize("dtype,tol", [(np.float64, 4e-14), (np.float32, 2.4e-5)])
@pytest.mark.parametrize("n", [64, 63, 41, 40, 33, 32, 2, 1])
@pytest.mark.parametrize("impl", [O.NAIVE, O.FAST])
def test_dct_is_orthonormal_dct2_dct3(dtype, tol, n, impl):
    rng = np.random.default_rng(n)
    a = rng.standard_normal(n).astype(dtype)
    b = O.dct_fwd(a, impl)
    np.testing.assert_allclose(b, sfft.dct(a.astype(np.float64), type=2, norm="ortho"), atol=tol)
    np.testing.assert_allclose(O.dct_inv(a, impl), sfft.dct(a.astype(np.float64), type=3, norm="ortho"), atol=tol)
    np.testing.assert_allclose(O.dct_inv(b, impl), a, atol=tol)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 2e-14), (np.float32, 1.2e-5)])
def test_fast_flow_matches_definition(dtype, tol):
    rng = np.random.default_rng(3)
    for _ in range(200):
        a = (rng.standard_normal(64) * 10 ** rng.uniform(-2, 1)).astype(dtype)
        s = max(1.0, np.abs(a).max())
        assert np.abs(O.dct_fwd(a, O.FAST) - O.dct_fwd(a, O.NAIVE)).max() < tol * s * 8
        assert np.abs(O.dct_inv(a, O.FAST) - O.dct_inv(a, O.NAIVE)).max() < tol * s * 8
    # linearity + impulse: DCT of e_j is column j of the orthonormal DCT matrix
    e = np.zeros(64, dtype); e[5] = 1
    k = np.arange(64)
    col = np.sqrt(2 / 64) * np.cos(np.pi * (2 * 5 + 1) * k / 128); col[0] /= np.sqrt(2)
    np.testing.assert_allclose(O.dct_fwd(e, O.FAST), col, atol=tol)


def test_tables_match_reference_formulae():
    as_, ax, ias, iax = O.dct_tables(64, np.float64)
    k = np.arange(64)
    ref = 2 * np.cos(-k * np.pi / 128) / np.sqrt(128.0); ref[0] /= np.sqrt(2.0)
    np.testing.assert_allclose(as_, ref, rtol=4e-16)
    np.testing.assert_allclose(iax, np.sin(k * np.pi / 128) * np.sqrt(128.0), rtol=4e-16, atol=1e-300)
    as5, ax5, _, _ = O.dct_tables(5, np.float64)      # odd length: no doubling (dct.c:43-52)
    assert abs(as5[1] - np.cos(-np.pi / 10) / np.sqrt(10.0)) < 1e-16


def test_bins_and_conv_table():
    bc = O.gen_bins(1e-3, np.float64)
    assert bc[0] == 0 and bc[1] == 2e-3 and bc[2] == -2e-3 and bc[253] == 127 * 2e-3 and bc[254] == -127 * 2e-3
    # conv_tbl (dctz-comp-lib.c:27-43) numbers the bins the way gen_bins centres
    # them: every binned coefficient lies within eb of its bin's centre, all 255
    # bins occur, and bin 0 is the one around zero.
    eb = 1e-3
    x = np.random.default_rng(5).uniform(-0.5, 0.5, 64 * 4000)
    x[0] = 5.0                                        # pins sf = 1
    c = O.compress(x, eb, O.EC, O.NAIVE, want_coef=True)
    assert c.sf == 1.0
    m = c.bin_index != 255
    assert np.abs(bc[c.bin_index[m]] - c.coef[m]).max() <= eb * (1 + 1e-12)
    assert set(np.unique(c.bin_index[m])) == set(range(255))
    assert np.abs(c.coef[c.bin_index == 0]).max() <= eb


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 128, 1000, 1001, 12960])
def test_roundtrip_error_bound_and_stream_consistency(dtype, mode, n):
    x = W.ragged(n, dtype)
    eb = 1e-3
    c = O.compress(x, eb, mode, O.FAST, want_coef=True)
    nblk = (n + 63) // 64
    # every block head is marked 255 (dctz-comp-lib.c:361); cnt = #255 - nblk
    assert np.all(c.bin_index[::64] == 255)
    assert int((c.bin_index == 255).sum()) - nblk == c.cnt
    # exceptions are emitted block-major, j ascending (dctz-comp-lib.c:478-544)
    idx = np.flatnonzero(c.bin_index == 255)
    idx = idx[idx % 64 != 0]
    if mode == O.EC:
        assert np.array_equal(c.ac_exact, c.coef[idx].astype(np.float32))
    r = O.decompress(c)
    orig = c.scaled * dtype(c.sf)
    # in-range coefficients are off by <= eb, exceptions by float truncation; the
    # orthonormal transform keeps the l2 error, so |err|_inf <= eb*sqrt(64)*sf (loose)
    assert np.abs(r.astype(np.float64) - orig.astype(np.float64)).max() <= 8.5 * eb * c.sf + 1e-6 * c.sf


def test_degenerate_and_error_paths():
    with pytest.raises(ValueError):
        O.compress(np.ones(64), 1e-7)                  # dctz-comp-lib.c:135-138
    c = O.compress(np.zeros(100), 1e-3)                # reference: sf = 0 -> NaN; ours: sf = 1
    assert c.sf == 1.0 and c.cnt == 0
    assert np.array_equal(O.decompress(c), np.zeros(100))


def test_psnr_restatement():
    x = np.array([0.0, 1.0, 2.0, 4.0]); r = x + np.array([0.1, -0.1, 0.0, 0.2])
    p = O.psnr(x, r)
    assert p["range"] == 4.0 and abs(p["maxdiff"] - 0.2) < 1e-15
    assert abs(p["psnr"] - 20 * np.log10(4.0 / np.sqrt((0.01 + 0.01 + 0.04) / 4))) < 1e-12


FFTW = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fftw_r2r_ref.json")))


@pytest.mark.parametrize("impl", [O.FAST, O.NAIVE])
@pytest.mark.parametrize("key,dtype,tol", [("f64", np.float64, 4e-15), ("f32", np.float32, 1e-6)])
def test_block_transform_against_fftw_generated_values(impl, key, dtype, tol):
    """The oracle's block transform against results of FFTW ITSELF: scipy's fftpack test-suite ships the outputs of FFTW's
    r2r transforms REDFT10 (DCT-II) and REDFT01 (DCT-III) on x = 0, 1, ..., n-1 (tests/golden/fftw_r2r_ref.json, made by
    tests/golden/scripts/make_fftw_r2r_fixture.py from scipy's data files; lengths 2 ... 64 -- all of them block lengths
    the codec can meet: 64, and the short last block).  Not the reference's own fixture -- it has none -- and not the
    complex-DFT route dct.c takes through FFTW, but values computed by the library the reference links, where scipy's own
    transform (pocketfft, the other independent check of this file) is not.
       REDFT10:  Y_k = 2 sum_j x_j cos(pi (j + 1/2) k / n)            -> orthonormal DCT-II  = sqrt(2/n) c_k Y_k / 2
       REDFT01:  Y_k = X_0 + 2 sum_{j>=1} X_j cos(pi j (k + 1/2) / n)  -> orthonormal DCT-III of (0, 1, ..., n-1)
                                                                         = sqrt(2/n) Y_k / 2   (X_0 = 0)"""
    for n in (2, 3, 4, 8, 12, 15, 16, 17, 32, 64):
        x = np.arange(n).astype(dtype)
        y2 = np.array([float(v) for v in FFTW[key]["dct_2_%d" % n]])
        c = np.ones(n)
        c[0] = 1.0 / np.sqrt(2.0)
        ref = y2 / 2.0 * np.sqrt(2.0 / n) * c
        got = O.dct_fwd(x, impl).astype(np.float64)
        assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), (n, "forward")
        y3 = np.array([float(v) for v in FFTW[key]["dct_3_%d" % n]])
        ref = y3 / 2.0 * np.sqrt(2.0 / n)
        got = O.dct_inv(x, impl).astype(np.float64)
        assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), (n, "inverse")
