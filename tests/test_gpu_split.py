"""k_compress_eo -- the chain's compress kernel with every block shared by a lane of an "even" and a lane of an "odd"
wavefront (dctz_amd/csrc/dctz_kernels_eo.hip) -- against the oracle and against k_compress: the same bytes on every stream,
the same header scalars.  Small arrays are forced onto the chain of kernels (set_one_launch(False)) so that the split
kernel runs at sizes the oracle finishes in seconds: one tile, a ragged last tile, a short last block, workgroups with
one tile and with several, dense and sparse exceptions (several staging rounds per tile), both modes, speculative
statistics on and off."""
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
    c.set_one_launch(False)
    yield c
    c.close()


def _dev(ctx, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)


def _same(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def _run(ctx, xd, eb, mode, split, coef=None):
    ctx.set_split(int(split))
    try:
        out, info = ctx.compress(xd, eb, mode, coef=coef)
        import torch
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in out.items()}, info
    finally:
        ctx.set_split(False)


def _check(ctx, x, eb, mode, want_coef=False):
    import torch
    xd = _dev(ctx, x)
    coef = torch.zeros_like(xd) if want_coef else None
    a, ia = _run(ctx, xd, eb, mode, 3, coef)          # split, AC_exact placed in the same pass (EC)
    l, il = _run(ctx, xd, eb, mode, 1)                # split, workgroup-local lists + k_compact_ac
    b, ib = _run(ctx, xd, eb, mode, 0)
    assert not (ia.flags & H.INFO_ONE_LAUNCH) and (ia.flags & H.INFO_SPLIT) and (il.flags & H.INFO_SPLIT) and not (ib.flags & H.INFO_SPLIT)
    if x.size >= 4096:
        assert bool(ia.flags & H.INFO_SINGLE_PASS) == (mode == O.EC) and not (il.flags & H.INFO_SINGLE_PASS)
    assert (il.cnt, il.sf) == (ib.cnt, ib.sf)
    assert torch.equal(l["bin_index"], b["bin_index"]) and torch.equal(l["dc"].view(torch.int32), b["dc"].view(torch.int32))
    assert torch.equal(l["ac_exact"][:il.cnt].view(torch.int32), b["ac_exact"][:ib.cnt].view(torch.int32))
    assert (ia.cnt, ia.sf, ia.max_abs, ia.min_abs, ia.nblk) == (ib.cnt, ib.sf, ib.max_abs, ib.min_abs, ib.nblk)
    assert torch.equal(a["bin_index"], b["bin_index"])
    assert torch.equal(a["dc"].view(torch.int32), b["dc"].view(torch.int32))
    assert torch.equal(a["ac_exact"][:ia.cnt].view(torch.int32), b["ac_exact"][:ib.cnt].view(torch.int32))
    assert list(ia.qtable) == list(ib.qtable) and list(ia.qtable_raw) == list(ib.qtable_raw)
    assert abs(ia.mean - ib.mean) <= 1e-9 * max(1.0, abs(ib.mean))
    c = O.compress(x, eb, mode, O.FAST, want_coef=want_coef)
    assert np.array_equal(xd.cpu().numpy(), x), "input must not be modified"
    assert ia.sf == c.sf and ia.cnt == c.cnt
    assert np.array_equal(a["bin_index"].cpu().numpy(), c.bin_index)
    assert _same(a["dc"].cpu().numpy(), c.dc)
    assert _same(a["ac_exact"][:c.cnt].cpu().numpy(), c.ac_exact)
    if mode == O.QT:
        assert _same(np.array(ia.qtable[:], dtype=x.dtype), c.qtable)
        assert _same(np.array(ia.qtable_raw[:], dtype=x.dtype), c.qtable_raw)
    if want_coef and mode == O.EC:
        assert _same(coef.cpu().numpy(), c.coef)
    return ia


SIZES = [64, 65, 4096, 4097, 4096 * 3 + 64 * 7, 4096 * 5 + 40, 4096 * 37 + 64 * 63 + 63, 1 << 20]


@pytest.mark.parametrize("mode", [O.EC, O.QT])
@pytest.mark.parametrize("n", SIZES)
def test_split_kernel_streams_bit_exact(ctx, mode, n):
    x = W.ragged(n, np.float64, scale=37.0)
    _check(ctx, x, 1e-3, mode, want_coef=(n <= 4096 * 5 + 40))


@pytest.mark.parametrize("mode", [O.EC, O.QT])
@pytest.mark.parametrize("eb", [1e-2, 1e-4, 1e-5, 1e-6])
def test_split_kernel_dense_and_sparse_exceptions(ctx, mode, eb):
    """eb 1e-2: hardly a coefficient stored exactly; 1e-5 / 1e-6: most of them (a tile's piece leaves in several rounds)."""
    x = W.ragged(4096 * 21 + 64 * 5 + 17, np.float64, scale=37.0)
    info = _check(ctx, x, eb, mode)
    if eb <= 1e-5:
        assert info.cnt > 0.5 * x.size


@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_split_kernel_c1_uniform_noise(ctx, mode):
    """Config 1 (2^20 uniform doubles, p = 0.915): every tile is dense."""
    info = _check(ctx, W.c1(), 1e-3, mode)
    assert info.cnt == 959612


@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_split_kernel_sf_one_and_speculation(ctx, mode):
    """sf == 1 (no scaling, dctz-comp-lib.c:193) and the speculative statistics (fused into the kernel: STATS) on a size
    that takes them."""
    x = (W.ragged(4096 * 9 + 5, np.float64, scale=0.6)).astype(np.float64)
    assert np.abs(x).max() < 1.0
    info = _check(ctx, x, 1e-3, mode)
    assert info.sf == 1.0 or info.sf == 0.1
    ctx.set_speculation(True, 1 << 18)
    try:
        y = W.ragged((1 << 20) + 4096 * 3 + 77, np.float64, scale=37.0)
        info = _check(ctx, y, 1e-3, mode)
        assert info.flags & H.INFO_STATS_FUSED
    finally:
        ctx.set_speculation(True, 1 << 22)


def test_split_kernel_special_values(ctx):
    """Signed zeros, a zero tile, huge and tiny magnitudes inside one array (the exact-division windows)."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(4096 * 6) * 10.0).astype(np.float64)
    x[:4096] = 0.0
    x[4096:4096 + 64] = -0.0
    x[3 * 4096 + 5] = 1e300
    x[3 * 4096 + 700] = 1e-310
    _check(ctx, x, 1e-3, O.EC)
    _check(ctx, x, 1e-3, O.QT)


def test_look_back_that_gives_up_falls_back_to_the_lists():
    """DCTZHIP_EO_LB_FAIL=1: one tile's look-back reports that it gave up; the call must come back with the lists' result --
    the same bytes -- say so in its flags, and stay on the lists for a while."""
    import os
    import torch
    import dctz_amd
    os.environ["DCTZHIP_EO_LB_FAIL"] = "1"
    try:
        c = dctz_amd.Context(0)
    finally:
        del os.environ["DCTZHIP_EO_LB_FAIL"]
    try:
        c.set_one_launch(False)
        x = W.ragged(4096 * 9 + 64 * 3 + 5, np.float64, scale=37.0)
        xd = torch.from_numpy(x).to(c.device)
        c.set_split(3)
        out, info = c.compress(xd, 1e-3, O.EC)
        torch.cuda.synchronize()
        assert info.flags & H.INFO_LB_FALLBACK and not (info.flags & H.INFO_SINGLE_PASS)
        ref = O.compress(x, 1e-3, O.EC, O.FAST)
        assert info.cnt == ref.cnt and np.array_equal(out["bin_index"].cpu().numpy(), ref.bin_index)
        assert _same(out["ac_exact"][:ref.cnt].cpu().numpy(), ref.ac_exact) and _same(out["dc"].cpu().numpy(), ref.dc)
        out2, info2 = c.compress(xd, 1e-3, O.EC)      # the pause: lists, no look-back at all
        torch.cuda.synchronize()
        assert (info2.flags & H.INFO_SPLIT) and not (info2.flags & (H.INFO_SINGLE_PASS | H.INFO_LB_FALLBACK))
        assert _same(out2["ac_exact"][:ref.cnt].cpu().numpy(), ref.ac_exact)
    finally:
        c.close()
