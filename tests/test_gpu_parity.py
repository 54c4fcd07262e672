"""Parity of the HIP path (through the C ABI) against the CPU oracle.

Bar: bit-exact on every stream -- bin_index (u8), DC and AC_exact (f32 bits),
cnt, sf, the QT table -- and on the reconstructed array (the kernels evaluate
the oracle's pinned expression tree, unfused)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import workloads as W
from dctz_amd import hip as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import dctz_amd
    c = dctz_amd.Context(0)
    yield c
    c.close()


def _dev(ctx, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def _same(a, b):
    return a.shape == b.shape and np.array_equal(_bits(a), _bits(b))


SIZES = [1, 2, 31, 63, 64, 65, 127, 128, 1000, 1001, 4096, 4097, 12960, 37024, 64 * 64 * 5 + 40]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", SIZES)
def test_dct_blocks_bit_exact(ctx, dtype, n):
    x = W.ragged(n, dtype)
    nblk = (n + 63) // 64
    for inverse in (False, True):
        y = ctx.dct_blocks(_dev(ctx, x), inverse=inverse).cpu().numpy()
        ref = np.empty_like(x)
        for b in range(nblk):
            sl = slice(64 * b, min(n, 64 * b + 64))
            ref[sl] = O.dct_inv(x[sl], O.FAST) if inverse else O.dct_fwd(x[sl], O.FAST)
        bad = np.flatnonzero(y.view(np.uint64 if dtype == np.float64 else np.uint32)
                             != ref.view(np.uint64 if dtype == np.float64 else np.uint32))
        assert bad.size == 0, (f"inverse={inverse} n={n}: {bad.size} mismatches, first at {bad[:8]} "
                               f"(block {bad[0] // 64}, j {bad[0] % 64}) maxdiff={np.abs(y - ref).max()}")


def _compress_both(ctx, x, eb, mode):
    import torch
    xd = _dev(ctx, x)
    coef = torch.empty_like(xd)
    scaled = torch.empty_like(xd)
    # Twice: without a scaled copy (the plain k_compress) and with one in a buffer of its own (the variant of k_compress
    # that writes x / sf back itself, dctz_kernels.hip: SC) -- the streams of the two must be the same bytes.
    plain, pinfo = ctx.compress(xd, eb, mode)
    keep = {k: plain[k].clone() for k in ("bin_index", "dc")}
    keep["ac_exact"] = plain["ac_exact"][:pinfo.cnt].clone()
    out, info = ctx.compress(xd, eb, mode, scaled=scaled, coef=coef)
    assert (pinfo.cnt, pinfo.sf) == (info.cnt, info.sf)
    assert torch.equal(keep["bin_index"], out["bin_index"]) and torch.equal(keep["dc"].view(torch.int32), out["dc"].view(torch.int32))
    assert torch.equal(keep["ac_exact"].view(torch.int32), out["ac_exact"][:info.cnt].view(torch.int32))
    # ... and a third time through the chain of kernels (arrays of this size take ONE kernel by default,
    # dctz_kernels_one.hip): the same bytes, the same header scalars
    assert pinfo.flags & H.INFO_ONE_LAUNCH and info.flags & H.INFO_ONE_LAUNCH
    ctx.set_one_launch(False)
    try:
        sc2, cf2 = torch.empty_like(xd), torch.empty_like(xd)
        chain, cinfo = ctx.compress(xd, eb, mode, scaled=sc2, coef=cf2)
    finally:
        ctx.set_one_launch(True)
    assert not (cinfo.flags & H.INFO_ONE_LAUNCH)
    assert (cinfo.cnt, cinfo.sf, cinfo.max_abs, cinfo.min_abs) == (info.cnt, info.sf, info.max_abs, info.min_abs)
    assert torch.equal(chain["bin_index"], out["bin_index"]) and torch.equal(chain["dc"].view(torch.int32), out["dc"].view(torch.int32))
    assert torch.equal(chain["ac_exact"][:info.cnt].view(torch.int32), out["ac_exact"][:info.cnt].view(torch.int32))
    assert torch.equal(sc2.view(torch.uint8), scaled.view(torch.uint8)) and torch.equal(cf2.view(torch.uint8), coef.view(torch.uint8))
    assert list(cinfo.qtable) == list(info.qtable) and list(cinfo.qtable_raw) == list(info.qtable_raw)
    c = O.compress(x, eb, mode, O.FAST, want_coef=True)
    return xd, out, info, coef, scaled, c


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
@pytest.mark.parametrize("n", SIZES)
def test_compress_streams_bit_exact(ctx, dtype, mode, n):
    x = W.ragged(n, dtype, scale=37.0)
    eb = 1e-3
    xd, out, info, coef, scaled, c = _compress_both(ctx, x, eb, mode)
    assert np.array_equal(xd.cpu().numpy(), x), "input must not be modified"
    assert info.sf == c.sf and info.nblk == (n + 63) // 64
    assert _same(scaled.cpu().numpy(), c.scaled)
    if mode == O.EC:   # (QT: the oracle's a_x holds the normalised values after pass 2)
        assert _same(coef.cpu().numpy(), c.coef)
    assert np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
    assert _same(out["dc"].cpu().numpy(), c.dc)
    assert info.cnt == c.cnt
    assert _same(out["ac_exact"][:c.cnt].cpu().numpy(), c.ac_exact)
    if mode == O.QT:
        assert _same(np.array(info.qtable[:], dtype=dtype), c.qtable)
        assert _same(np.array(info.qtable_raw[:], dtype=dtype), c.qtable_raw)
    # mean: device summation order differs from util.c's serial loop
    assert abs(info.mean - c.mean) <= 1e-5 * max(1.0, abs(c.mean))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
@pytest.mark.parametrize("n", SIZES)
def test_decompress_bit_exact(ctx, dtype, mode, n):
    import torch
    x = W.ragged(n, dtype, scale=37.0)
    eb = 1e-3
    c = O.compress(x, eb, mode, O.FAST)
    ref = O.decompress(c, O.FAST)
    out = {"bin_index": _dev(ctx, c.bin_index), "dc": _dev(ctx, c.dc),
           "ac_exact": _dev(ctx, c.ac_exact if c.cnt else np.zeros(4, np.float32))}
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    r = ctx.decompress(out, c.cnt, n, tdt, eb, c.sf, mode, qtable=c.qtable).cpu().numpy()
    assert _same(r, ref), f"maxdiff={np.abs(r - ref).max()}"
    ctx.set_one_launch(False)                          # ... and through the chain of kernels
    try:
        r2 = ctx.decompress(out, c.cnt, n, tdt, eb, c.sf, mode, qtable=c.qtable).cpu().numpy()
    finally:
        ctx.set_one_launch(True)
    assert _same(r2, ref)


@pytest.mark.parametrize("eb", [1e-3, 1e-4, 1e-5, 1e-6])
def test_c1_known_answer_on_gpu(ctx, eb):
    """Config 1 (2^20 uniform doubles): at eb = 1e-3 the survey's recorded outputs
    of the reference (cnt, PSNR, max|err|) must come out of the HIP path."""
    import json, os, torch
    x = W.c1()
    xd = _dev(ctx, x)
    out, info = ctx.compress(xd, eb, O.EC)
    c = O.compress(x, eb, O.EC, O.FAST)
    assert info.sf == c.sf and info.cnt == c.cnt
    assert np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
    assert _same(out["ac_exact"][:c.cnt].cpu().numpy(), c.ac_exact)
    r = ctx.decompress(out, info.cnt, x.size, torch.float64, eb, info.sf, O.EC).cpu().numpy()
    assert _same(r, O.decompress(c, O.FAST))
    if eb == 1e-3:
        ka = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_known_answers.json")))["C1_ec"]
        p = O.psnr((x / info.sf) * info.sf, r)
        assert info.cnt == ka["cnt"] and info.sf == ka["sf"]
        assert abs(p["psnr"] - ka["psnr"]) / ka["psnr"] < 1e-6
        assert abs(p["maxdiff"] - ka["maxerr"]) / ka["maxerr"] < 1e-6


def test_c2_fp32_field(ctx):
    import torch
    f = W.c2()
    for mode in (O.EC, O.QT):
        out, info = ctx.compress(_dev(ctx, f), 1e-4, mode)
        c = O.compress(f, 1e-4, mode, O.FAST)
        assert info.sf == c.sf and info.cnt == c.cnt
        assert np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
        assert _same(out["dc"].cpu().numpy(), c.dc)
        assert _same(out["ac_exact"][:c.cnt].cpu().numpy(), c.ac_exact)
        r = ctx.decompress(out, info.cnt, f.size, torch.float32, 1e-4, info.sf, mode,
                           qtable=np.array(info.qtable[:])).cpu().numpy()
        assert _same(r, O.decompress(c, O.FAST))
        assert round(O.psnr(f, r)["psnr"], 2) == 93.63     # survey known answer (C2-like)


def test_error_paths(ctx):
    import dctz_amd, torch
    x = torch.ones(128, dtype=torch.float64, device=ctx.device)
    with pytest.raises(dctz_amd.DctzHipError, match="ERROR BOUND"):
        ctx.compress(x, 1e-7)                            # dctz-comp-lib.c:135-138
    out, info = ctx.compress(torch.zeros(100, dtype=torch.float64, device=ctx.device), 1e-3)
    assert info.sf == 1.0 and info.cnt == 0              # documented deviation (sf = 1)
    # a bin_index that flags more exceptions than the stream carries is refused
    c = O.compress(W.ragged(1000, np.float64, scale=37.0), 1e-3)
    outs = {"bin_index": _dev(ctx, c.bin_index), "dc": _dev(ctx, c.dc), "ac_exact": _dev(ctx, c.ac_exact)}
    with pytest.raises(dctz_amd.DctzHipError):
        ctx.decompress(outs, max(c.cnt - 1, 0), 1000, torch.float64, 1e-3, c.sf)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fast_division_is_exact(ctx, dtype):
    """The kernels' hoisted-reciprocal division == the compiler's IEEE division ==
    numpy's, bit for bit: random magnitudes over the whole exponent range, values
    straddling the fast-path window, zeros of both signs, denormals, inf, nan, and
    quotients sitting next to rounding midpoints."""
    import torch
    rng = np.random.default_rng(99)
    info = np.finfo(dtype)
    divisors = [0.1, 10.0, 100.0, 1e-3, 2e-3, 2e-4, 2e-5, 2e-6, 1e5, 3.0, 0.002 * (1 + 2 ** -20), 1e30 if dtype == np.float64 else 1e9,
                1e-30 if dtype == np.float64 else 1e-9, 1e300 if dtype == np.float64 else 1e30]
    m = 1 << 20
    mant = rng.uniform(1, 2, m)
    expo = rng.integers(info.minexp - 10, info.maxexp, m)
    wide = np.ldexp(mant, expo).astype(dtype) * rng.choice([-1.0, 1.0], m).astype(dtype)
    narrow = (rng.standard_normal(m) * 10 ** rng.uniform(-3, 3, m)).astype(dtype)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, info.tiny, -info.tiny, info.max, -info.max,
                        info.smallest_subnormal, 1.0, -1.0], dtype=dtype)
    for d in divisors:
        dd = dtype(d)
        # x = RN(d * (k + 1/2 ulp-ish)): quotients next to the midpoints between floats
        k = rng.uniform(1, 2, 1 << 16).astype(dtype)
        mid = (k + np.spacing(k) / 2).astype(np.float64) * np.float64(dd)
        near = np.concatenate([np.nextafter(mid.astype(dtype), dtype(np.inf)), mid.astype(dtype),
                               np.nextafter(mid.astype(dtype), dtype(-np.inf))])
        x = np.concatenate([wide, narrow, special, near])
        fast, ref = ctx.debug_divide(_dev(ctx, x), float(dd))
        fast, ref = fast.cpu().numpy(), ref.cpu().numpy()
        with np.errstate(all="ignore"):
            host = (x / dd).astype(dtype)
        nan = np.isnan(host)
        assert np.array_equal(np.isnan(fast), nan) and np.array_equal(np.isnan(ref), nan)
        assert np.array_equal(fast[~nan].view(np.uint8), ref[~nan].view(np.uint8)), f"fast != device '/' for d={d}"
        assert np.array_equal(ref[~nan].view(np.uint8), host[~nan].view(np.uint8)), f"device '/' != IEEE for d={d}"


# ---- speculative fused statistics (include/dctz_hip.h, DCTZHIP_INFO_*) -------------------
def _sampled_chunk(g, group=64):
    """Chunk index k_stats_sample reads in group g (dctz_kernels.hip: hashed position)."""
    return g * group + (((g * 2654435761) & 0xFFFFFFFF) >> 8) % group


def _unsampled_element(dtype, group_index, group=64):
    chunk_elems = 256 * (16 // np.dtype(dtype).itemsize)          # SWG threads x one 16-byte vector
    g0 = _sampled_chunk(group_index, group)
    c = group_index * group + (0 if g0 != group_index * group else 1)
    assert c != g0
    return c * chunk_elems + 17


@pytest.fixture()
def spec_ctx():
    import dctz_amd
    c = dctz_amd.Context(0)
    c.set_speculation(True, 1 << 18)
    c.set_one_launch(False)                 # (these tests are about the chain's speculative statistics)
    yield c
    c.close()


def _check_against_oracle(ctx, x, eb, mode, out, info):
    c = O.compress(x, eb, mode, O.FAST)
    assert info.sf == c.sf and info.cnt == c.cnt
    assert info.max_abs == float(np.abs(x).max()) and info.min_abs == float(np.abs(x).min())
    assert np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
    assert _same(out["dc"].cpu().numpy(), c.dc)
    assert _same(out["ac_exact"][:c.cnt].cpu().numpy(), c.ac_exact)
    assert abs(info.mean - c.mean) <= 1e-5 * max(1.0, abs(c.mean))
    if mode == O.QT:
        assert _same(np.array(info.qtable[:], dtype=x.dtype), c.qtable)
    return c


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
@pytest.mark.parametrize("n", [1 << 20, (1 << 20) + 64 * 7 + 40, 300000])
def test_speculation_hit_is_bit_exact(spec_ctx, dtype, mode, n):
    import dctz_amd
    x = W.ragged(n, dtype, scale=37.0)
    out, info = spec_ctx.compress(_dev(spec_ctx, x), 1e-3, mode)
    assert info.flags == dctz_amd.hip.INFO_STATS_FUSED, "smooth data: the sampled guess must verify"
    _check_against_oracle(spec_ctx, x, 1e-3, mode, out, info)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_speculation_wrong_decade_is_detected_and_rerun(spec_ctx, dtype, mode):
    """A spike the sample cannot see moves max|x| into the next decade: the guess of sf is
    wrong, the fused statistics say so, and the second run gives the reference's streams."""
    import dctz_amd
    n = 1 << 20
    x = W.ragged(n, dtype, scale=37.0)
    x[_unsampled_element(dtype, 5)] = 4321.0
    out, info = spec_ctx.compress(_dev(spec_ctx, x), 1e-3, mode)
    assert info.flags == dctz_amd.hip.INFO_RESPUN
    c = _check_against_oracle(spec_ctx, x, 1e-3, mode, out, info)
    assert c.sf == 1000.0
    # cool-down: the next calls take the plain statistics pass, then speculation resumes
    for i in range(9):
        out, info = spec_ctx.compress(_dev(spec_ctx, x), 1e-3, mode)
        assert info.flags == (0 if i < 8 else dctz_amd.hip.INFO_RESPUN), i
        _check_against_oracle(spec_ctx, x, 1e-3, mode, out, info)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_speculation_window_violation_is_detected(spec_ctx, dtype):
    """An exact zero the sample cannot see: the guessed 'no per-element division test' level
    would be unproven, so the call must fall back (zeros divide exactly either way, but the
    proof obligation is on min|x|)."""
    import dctz_amd
    n = 1 << 20
    x = W.ragged(n, dtype, scale=37.0)
    x[x == 0] = 1.0
    x[_unsampled_element(dtype, 9)] = 0.0
    tiny = np.finfo(dtype).tiny
    x[_unsampled_element(dtype, 11)] = tiny * 4          # far below FastDiv's window
    out, info = spec_ctx.compress(_dev(spec_ctx, x), 1e-3, O.EC)
    assert info.flags == dctz_amd.hip.INFO_RESPUN
    _check_against_oracle(spec_ctx, x, 1e-3, O.EC, out, info)


def test_in_place_scaling_under_speculation_and_small_inputs(spec_ctx):
    """d_scaled may alias d_in (the reference's in-place division, dctz-comp-lib.c:193-216): such a call leaves the
    speculative path (k_compress writes x / sf over the input itself, which needs the verified sf), so a spike the sample
    cannot see does no harm; and everything else about the call is what the oracle says."""
    import torch
    x = W.ragged(1 << 20, np.float64, scale=37.0)
    c = O.compress(x, 1e-3, O.EC, O.FAST)
    for spike in (False, True):
        z = x.copy()
        if spike:
            z[12345] = 4.0e4                                         # a sample would not see it (wrong decade)
        cz = O.compress(z, 1e-3, O.EC, O.FAST)
        xd = _dev(spec_ctx, z)
        out, info = spec_ctx.compress(xd, 1e-3, O.EC, scaled=xd)    # d_scaled aliases d_in
        if os.environ.get("DCTZHIP_FUSE_SCALED", "1") != "0":
            assert info.flags == 0                                  # neither speculated nor run twice
        else:                                                       # (the separate k_scale pass: written last, with the verified sf)
            assert info.flags in ((H.INFO_RESPUN,) if spike else (H.INFO_STATS_FUSED,))
        assert info.sf == cz.sf and info.cnt == cz.cnt
        assert _same(xd.cpu().numpy(), cz.scaled)
        assert np.array_equal(out["bin_index"].cpu().numpy(), cz.bin_index)
    y = W.ragged(1 << 16, np.float64, scale=37.0)                 # below the threshold
    out, info = spec_ctx.compress(_dev(spec_ctx, y), 1e-3, O.EC)
    assert info.flags == 0
    spec_ctx.set_speculation(False)
    out, info = spec_ctx.compress(_dev(spec_ctx, x), 1e-3, O.EC)
    assert info.flags == 0 and info.cnt == c.cnt


@pytest.mark.parametrize("one", [True, False])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
@pytest.mark.parametrize("n", [64 * 64 * 3 + 64 * 9 + 21, 64 * 5 + 7, 1 << 20, 40])
@pytest.mark.parametrize("sf_one", [False, True])
def test_in_place_scaling_every_layout(ctx, one, dtype, mode, n, sf_one):
    """d_scaled == d_in (what the reference does to its caller's array, dctz-comp-lib.c:193-216) for both element types and
    modes, whole tiles, a partial last tile, a remainder block, an array shorter than a block, and sf == 1 -- through the
    one-launch kernel and through the chain (only fp64 / EC / whole tiles were pinned)."""
    x = W.ragged(n, dtype, scale=(5.0 if sf_one else 37.0))
    c = O.compress(x, 1e-3, mode, O.FAST)
    assert (c.sf == 1.0) == sf_one
    xd = _dev(ctx, x)
    ctx.set_one_launch(one)
    try:
        out, info = ctx.compress(xd, 1e-3, mode, scaled=xd)
    finally:
        ctx.set_one_launch(True)
    # (round 5, ADVICE r4: a call whose scaled copy goes over its input always takes the chain of kernels -- the one-launch
    # kernel stores x / sf before the sweeps that can still give up, and a half-divided input cannot be run again)
    assert not (info.flags & H.INFO_ONE_LAUNCH)
    assert info.sf == c.sf and info.cnt == c.cnt
    assert _same(xd.cpu().numpy(), c.scaled)
    assert np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
    assert _same(out["dc"].cpu().numpy(), c.dc)
    assert _same(out["ac_exact"][:c.cnt].cpu().numpy(), c.ac_exact)
    if mode == O.QT:
        assert _same(np.array(info.qtable[:], dtype=dtype), c.qtable)


@pytest.mark.parametrize("forced", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_one_launch_wrong_guess_of_the_decade_is_replayed(dtype, mode, forced, monkeypatch):
    """k_compress_one scales on a guess of the array's decade (own tiles + a 1024-element sample) and verifies it against
    the board afterwards; a wrong guess runs the tile again from its image in LDS.  forced: the library makes every first
    guess wrong (DCTZHIP_ONE_BADGUESS); else a spike between the sample's positions, which only its own workgroup sees."""
    import dctz_amd
    monkeypatch.setenv("DCTZHIP_ONE_BADGUESS", "1" if forced else "3")    # (3: guess whatever the grid, not only from 128 workgroups on)
    c = dctz_amd.Context(0)
    try:
        for n in ((1 << 20) + 64 * 3 + 9, 64 * 64 * 7 + 5):
            x = W.ragged(n, dtype, scale=37.0)
            if not forced:
                x[12345 + 64 * 7] = 4.0e4              # (the sample reads 64 whole blocks: block k nfull / 64)
            ref = O.compress(x, 1e-3, mode, O.FAST)
            out, info = c.compress(_dev(c, x), 1e-3, mode)
            assert info.flags & H.INFO_ONE_LAUNCH
            assert info.sf == ref.sf and info.cnt == ref.cnt and info.max_abs == float(np.abs(x).max())
            assert np.array_equal(out["bin_index"].cpu().numpy(), ref.bin_index)
            assert _same(out["dc"].cpu().numpy(), ref.dc)
            assert _same(out["ac_exact"][:ref.cnt].cpu().numpy(), ref.ac_exact)
            if mode == O.QT:
                assert _same(np.array(info.qtable[:], dtype=dtype), ref.qtable)
    finally:
        c.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_multi_tile_workgroup_ranges_bit_exact(ctx, dtype, mode):
    """More tiles than workgroups (every workgroup walks a RANGE of tiles and flushes tile k while
    it builds tile k+1).  Regression test: a reconstruction whose store data registers were reused
    by the next LDS reads came out corrupted in a few blocks per million, QT mode only, and not on
    every run -- hence three repetitions against the oracle, bit for bit."""
    import torch
    n = (1 << 23) + 64 * 9 + 21
    x = W.ragged(n, dtype, scale=37.0)
    c = O.compress(x, 1e-3, mode, O.FAST)
    ref = O.decompress(c, O.FAST)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    out, info = ctx.compress(_dev(ctx, x), 1e-3, mode)
    assert info.cnt == c.cnt and np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
    assert _same(out["ac_exact"][:c.cnt].cpu().numpy(), c.ac_exact)
    for _ in range(3):
        r = ctx.decompress(out, info.cnt, n, tdt, 1e-3, info.sf, mode, qtable=np.array(info.qtable[:])).cpu().numpy()
        it = np.uint64 if dtype == np.float64 else np.uint32
        bad = np.flatnonzero(r.view(it) != ref.view(it))
        assert bad.size == 0, f"{bad.size} mismatches, first {bad[:8]} (tiles {np.unique(bad // 1024)[:6]})"


def _special(kind, n, dtype):
    """Inputs that take the less-travelled branches: division levels 1 / 0 (zeros, extreme
    exponents), sf == 1 (no scaling kernels), constant / sign-uniform data, tight and loose bounds."""
    rng = np.random.default_rng(99)
    base = W.ragged(n, np.float64, scale=37.0)
    if kind == "zeros_sprinkled":              # land-mask-like: exact zeros -> per-element window test
        base[rng.random(n) < 0.3] = 0.0
    elif kind == "constant":
        base[:] = 2.5
    elif kind == "sf_is_one":                  # max|x| in (1, 10]: sf == 1, SCALE = false variants
        base = base / 37.0 * 5.0
    elif kind == "negative":
        base = -np.abs(base) - 1.0
    elif kind == "huge":                       # near the top of the exponent range (fp32: 1e30)
        base = base * (1e290 if dtype == np.float64 else 1e28)
    elif kind == "tiny":                       # below FastDiv's window: plain IEEE division path
        base = base * (1e-290 if dtype == np.float64 else 1e-30)
    elif kind == "spiky":                      # heavy tails: most coefficients out of range
        base = base + 200.0 * rng.standard_cauchy(n).clip(-1e3, 1e3)
    return base.astype(dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
@pytest.mark.parametrize("kind,eb", [("zeros_sprinkled", 1e-3), ("constant", 1e-3), ("sf_is_one", 1e-3), ("negative", 1e-4),
                                     ("huge", 1e-3), ("tiny", 1e-3), ("spiky", 1e-2), ("spiky", 1e-6)])
def test_special_inputs_bit_exact(ctx, dtype, mode, kind, eb):
    import torch
    n = (1 << 22) + 64 * 11 + 5                # multi-tile workgroup ranges + a short last block
    x = _special(kind, n, dtype)
    c = O.compress(x, eb, mode, O.FAST)
    ref = O.decompress(c, O.FAST)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    out, info = ctx.compress(_dev(ctx, x), eb, mode)
    assert info.sf == c.sf and info.cnt == c.cnt
    assert np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
    assert _same(out["dc"].cpu().numpy(), c.dc)
    assert _same(out["ac_exact"][:c.cnt].cpu().numpy(), c.ac_exact)
    if mode == O.QT:
        assert _same(np.array(info.qtable[:], dtype=dtype), c.qtable)
    r = ctx.decompress(out, info.cnt, n, tdt, eb, info.sf, mode, qtable=np.array(info.qtable[:])).cpu().numpy()
    assert _same(r, ref), f"maxdiff={np.abs(r - ref).max()}"


@pytest.mark.parametrize("hit", [True, False])
def test_speculation_with_scaled_and_coefficient_taps(spec_ctx, hit):
    """The optional outputs (x/sf and the raw coefficients) must be right on the speculative path
    too -- after a verified guess and after a re-run with the true statistics."""
    import torch, dctz_amd
    n = (1 << 20) + 64 * 3 + 17
    x = W.ragged(n, np.float64, scale=37.0)
    if not hit:
        x[_unsampled_element(np.float64, 7)] = -7777.0
    xd = _dev(spec_ctx, x)
    scaled, coef = torch.empty_like(xd), torch.empty_like(xd)
    out, info = spec_ctx.compress(xd, 1e-3, O.EC, scaled=scaled, coef=coef)
    assert info.flags == (dctz_amd.hip.INFO_STATS_FUSED if hit else dctz_amd.hip.INFO_RESPUN)
    c = O.compress(x, 1e-3, O.EC, O.FAST, want_coef=True)
    assert info.sf == c.sf and info.cnt == c.cnt
    assert _same(scaled.cpu().numpy(), c.scaled) and _same(coef.cpu().numpy(), c.coef)
    assert np.array_equal(xd.cpu().numpy(), x), "input must not be modified"


# ---- fuzz: random shapes / magnitudes / bounds, HIP path vs oracle, bit for bit ------------------
from hypothesis import HealthCheck, given, settings, strategies as st   # noqa: E402


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2**31), n=st.integers(1, 400000), log_amp=st.floats(-6, 8), noise=st.sampled_from([0.0, 1e-5, 0.02, 1.0]),
       eb=st.sampled_from([1e-2, 1e-3, 1e-4, 1e-6]), mode=st.sampled_from([O.EC, O.QT]), dtype=st.sampled_from([np.float64, np.float32]),
       zeros=st.booleans())
def test_fuzz_against_oracle(ctx, seed, n, log_amp, noise, eb, mode, dtype, zeros):
    import torch
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 41.0
    amp = 10.0 ** log_amp
    x = amp * (np.sin(t) + 0.3 * np.cos(4.7 * t)) + noise * amp * rng.standard_normal(n)
    if zeros:
        x[rng.random(n) < 0.1] = 0.0
    x = x.astype(dtype)
    c = O.compress(x, eb, mode, O.FAST)
    out, info = ctx.compress(_dev(ctx, x), eb, mode)
    assert info.sf == c.sf and info.cnt == c.cnt
    assert np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
    assert _same(out["dc"].cpu().numpy(), c.dc)
    assert _same(out["ac_exact"][:c.cnt].cpu().numpy(), c.ac_exact)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    r = ctx.decompress(out, info.cnt, n, tdt, eb, info.sf, mode, qtable=np.array(info.qtable[:])).cpu().numpy()
    assert _same(r, O.decompress(c, O.FAST))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("part", [64 * 64 * 3, 64 * 1000, 64 * 64 * 40 + 64 * 5])
def test_parts_with_the_arrays_statistics_are_the_one_call(ctx, dtype, part):
    """dctzhip_compress_part: an array compressed part by part -- parts of whole blocks, every part scaled by the ARRAY's
    scaling factor (max|x| / min|x| given by the caller), AC_exact appended behind the parts in front -- gives the bytes of
    the one call and of the oracle; the parts' sums add up to the array's (util.c:22: from the second element on).  The
    array's largest value sits in the LAST part: a part scaled by its own statistics would differ."""
    import torch
    n = part * 3 + 64 * 17 + 29
    x = W.ragged(n, dtype, scale=3.0)
    x[n - 50] = -4321.0
    eb = 1e-3
    c = O.compress(x, eb, O.EC, O.FAST)
    xd = _dev(ctx, x)
    out = ctx.alloc_outputs(n, xd.dtype)
    for k in out:
        out[k].zero_()
    mx, mn = float(np.abs(x).max()), float(np.abs(x).min())
    S, total, lo = 0, 0.0, 0
    while lo < n:
        ne = min(part, n - lo)
        cnt, st, sf = ctx.compress_part(xd[lo:lo + ne], eb, mx, mn, out, lo, S)
        assert sf == c.sf
        assert st[0] == float(np.abs(x[lo:lo + ne]).max()) and st[1] == float(np.abs(x[lo:lo + ne]).min())
        total += st[2] + (float(x[lo]) if lo else 0.0)
        S += cnt
        lo += ne
    assert S == c.cnt
    assert np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
    assert _same(out["dc"].cpu().numpy(), c.dc)
    assert _same(out["ac_exact"][:S].cpu().numpy(), c.ac_exact)
    assert abs(total / n - c.mean) <= 1e-5 * max(1.0, abs(c.mean))
    # statistics that are not the array's are refused
    with pytest.raises(H.DctzHipError):
        ctx.compress_part(xd[n - 64 * 17 - 29:], eb, 10.0, mn, out, n - 64 * 17 - 29, 0)


@pytest.mark.parametrize("key,dtype,tol", [("f64", np.float64, 4e-15), ("f32", np.float32, 1e-6)])
def test_block_transform_on_the_gpu_against_fftw_generated_values(ctx, key, dtype, tol):
    """dctzhip_dct_blocks against results of FFTW itself (tests/golden/fftw_r2r_ref.json: FFTW's REDFT10 / REDFT01 on
    0, 1, ..., n-1, from scipy's fftpack test data; tests/test_oracle.py has the scaling): the HIP path, not only the
    oracle, agrees with the library the reference links to rounding, for the full block and for short last blocks."""
    import json
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fftw_r2r_ref.json")))
    for n in (2, 3, 4, 8, 12, 15, 16, 17, 32, 64):
        x = np.arange(n).astype(dtype)
        y2 = np.array([float(v) for v in ref[key]["dct_2_%d" % n]])
        c = np.ones(n)
        c[0] = 1.0 / np.sqrt(2.0)
        want = y2 / 2.0 * np.sqrt(2.0 / n) * c
        got = ctx.dct_blocks(_dev(ctx, x), inverse=False).cpu().numpy().astype(np.float64)
        assert np.abs(got - want).max() <= tol * np.abs(want).max(), (n, "forward")
        y3 = np.array([float(v) for v in ref[key]["dct_3_%d" % n]])
        want = y3 / 2.0 * np.sqrt(2.0 / n)
        got = ctx.dct_blocks(_dev(ctx, x), inverse=True).cpu().numpy().astype(np.float64)
        assert np.abs(got - want).max() <= tol * np.abs(want).max(), (n, "inverse")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_under_run_is_refused_on_the_large_array_path(ctx, dtype):
    """A header that promises fewer exact coefficients than bin_index flags (a damaged or truncated container) through the
    persistent decode kernels -- for fp64 EC the tile-interleaved k_decompress_il, whose workgroups address AC_exact through a
    descriptor per tile: the call is refused, nothing is read beyond ac_count (the descriptors end there), and the context
    decodes the intact stream bit for bit afterwards."""
    import dctz_amd, torch
    n = (1 << 23) + 64 * 5 + 3
    x = W.ragged(n, dtype, scale=37.0)
    xd = _dev(ctx, x)
    out, info = ctx.compress(xd, 1e-3, O.EC)
    assert not (info.flags & H.INFO_ONE_LAUNCH)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    good = ctx.decompress(out, info.cnt, n, tdt, 1e-3, info.sf, O.EC).cpu().numpy()
    for short in (1, 5000, info.cnt):
        with pytest.raises(dctz_amd.DctzHipError):
            ctx.decompress(out, info.cnt - short, n, tdt, 1e-3, info.sf, O.EC)
    again = ctx.decompress(out, info.cnt, n, tdt, 1e-3, info.sf, O.EC).cpu().numpy()
    assert _same(good, again)
    ref = O.decompress(O.compress(x, 1e-3, O.EC, O.FAST), O.FAST)
    assert _same(good, ref)
