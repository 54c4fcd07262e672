"""Batches of arrays (dctzhip_compress_batch / dctzhip_decompress_batch; VERDICT r2 "Next round" 1b).

The reference's own workloads are lists of small arrays compressed one call -- one process -- each
(tests/test-dctz.sh:13-56 over tests/list-msst19.txt:1-6, tests/list-CESM-ATM-tylor.txt:1-5).  The batch entry points run
such a list through one launch sequence per element type; nothing in the reference couples two arrays
(dctz-comp-lib.c:186 onwards: own calc_data_stat, sf, bin ranges, tot_AC_exact_count, QT table), so the bar is: every
array of a batch bit-identical to its own single-array call AND to the oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import workloads as W

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


def _tdt(a):
    import torch
    return torch.float64 if a.dtype == np.float64 else torch.float32


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def _check_compress(x, eb, mode, out, info, scaled=None):
    """Streams and header scalars of one array of a batch against the oracle's pinned flow."""
    c = O.compress(x, eb, mode, O.FAST)
    assert info.sf == c.sf and info.cnt == c.cnt, (x.size, x.dtype, eb, info.sf, c.sf, info.cnt, c.cnt)
    assert info.nblk == (x.size + 63) // 64
    assert info.max_abs == c.stats.max and info.min_abs == c.stats.min
    # (tree-order sum on the device; the reference adds serially in the data type, util.c:18-28 / :31-41, so an fp32
    # array is compared with the exact sum instead of the oracle's float accumulation)
    want = c.mean if x.dtype == np.float64 else float(x[1:].astype(np.float64).sum()) / x.size
    assert abs(info.mean - want) <= 1e-6 * max(info.max_abs, 1e-300) + 1e-300
    assert np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
    assert np.array_equal(_bits(out["dc"].cpu().numpy()), _bits(c.dc))
    assert np.array_equal(_bits(out["ac_exact"][:c.cnt].cpu().numpy()), _bits(c.ac_exact))
    if mode == O.QT:
        assert np.array_equal(_bits(np.array(info.qtable[:]).astype(x.dtype)), _bits(c.qtable))
        assert np.array_equal(_bits(np.array(info.qtable_raw[1:]).astype(x.dtype)), _bits(c.qtable_raw[1:]))
    if scaled is not None:
        assert np.array_equal(_bits(scaled.cpu().numpy()), _bits(c.scaled))
    return c


def _c5_list():
    """Config C5: the six list-msst19 lengths (fp64) x eb 1e-3 .. 1e-6, plus the CESM-sized fp32 field (C2)."""
    xs, ebs = [], []
    for i, n in enumerate(W.MSST19_LENGTHS):
        for eb in (1e-3, 1e-4, 1e-5, 1e-6):
            xs.append(W.c5_fp64(n, 100 + i))
            ebs.append(eb)
    xs.append(W.c2())
    ebs.append(1e-4)
    return xs, ebs


@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_c5_batch_bit_identical_to_single_calls_and_oracle(ctx, mode):
    import torch
    xs, ebs = _c5_list()
    xd = [_dev(ctx, x) for x in xs]
    scaled = [torch.empty_like(t) for t in xd]
    outs, infos, _ = ctx.compress_batch(xd, ebs, mode, scaled=scaled)
    torch.cuda.synchronize()
    cs = []
    for x, t, eb, out, info, sc in zip(xs, xd, ebs, outs, infos, scaled):
        cs.append(_check_compress(x, eb, mode, out, info, sc))
        o1, i1 = ctx.compress(t, eb, mode)                           # the array's own call
        assert i1.cnt == info.cnt and i1.sf == info.sf and abs(i1.mean - info.mean) <= 1e-12 * info.max_abs      # (tree-order sums)
        for k in ("bin_index", "dc"):
            assert torch.equal(o1[k], out[k])
        assert torch.equal(o1["ac_exact"][:info.cnt], out["ac_exact"][:info.cnt])
    # and back, as a batch
    dsts, status, _ = ctx.decompress_batch(outs, [i.cnt for i in infos], [x.size for x in xs], [_tdt(x) for x in xs], ebs,
                                           [i.sf for i in infos], mode, qtables=[np.array(i.qtable[:]) for i in infos])
    torch.cuda.synchronize()
    assert all(s == 0 for s in status)
    for c, d in zip(cs, dsts):
        assert np.array_equal(_bits(d.cpu().numpy()), _bits(O.decompress(c, O.FAST)))


@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_ragged_mixed_batch(ctx, mode):
    """Every shape a single call handles, side by side: arrays shorter than a block, exact blocks, odd / even remainders,
    several tiles, a flat-zero array, both element types, different bounds and amplitudes."""
    import torch
    rng = np.random.default_rng(31)
    sizes = [1, 2, 63, 64, 65, 127, 128, 1000, 1001, 4096, 4097, 4096 * 3 + 64 * 5 + 33, 12960, 37024, 4096 * 40 + 7]
    xs, ebs = [], []
    for i, n in enumerate(sizes):
        dt = np.float64 if i % 2 == 0 else np.float32
        xs.append(W.ragged(n, dt, seed=i, scale=float(10.0 ** rng.integers(-3, 4))))
        ebs.append(float(rng.choice([1e-2, 1e-3, 1e-4, 1e-5])))
    xs.append(np.zeros(777, np.float64)); ebs.append(1e-3)
    xs.append(rng.random(64 * 70 + 5).astype(np.float32)); ebs.append(1e-3)      # noise: nearly everything stored exactly
    xd = [_dev(ctx, x) for x in xs]
    outs, infos, _ = ctx.compress_batch(xd, ebs, mode)
    torch.cuda.synchronize()
    cs = [_check_compress(x, eb, mode, out, info) for x, eb, out, info in zip(xs, ebs, outs, infos)]
    dsts, status, _ = ctx.decompress_batch(outs, [c.cnt for c in cs], [x.size for x in xs], [_tdt(x) for x in xs], ebs,
                                           [c.sf for c in cs], mode, qtables=[c.qtable for c in cs])
    torch.cuda.synchronize()
    assert all(s == 0 for s in status)
    for c, d in zip(cs, dsts):
        assert np.array_equal(_bits(d.cpu().numpy()), _bits(O.decompress(c, O.FAST)))


def test_batch_in_place_scaling_and_repeat(ctx):
    """d_scaled may alias d_in (the reference scales the caller's buffer in place, dctz-comp-lib.c:193-216); a prepared
    batch can be issued again."""
    import torch
    xs = [W.c5_fp64(n, 7 + i) for i, n in enumerate(W.MSST19_LENGTHS)] + [W.ragged(5000, np.float32, scale=420.0)]
    xd = [_dev(ctx, x) for x in xs]
    keep = [t.clone() for t in xd]
    outs, infos, prep = ctx.compress_batch(xd, 1e-3, O.EC, scaled=xd)
    torch.cuda.synchronize()
    for x, t, out, info in zip(xs, xd, outs, infos):
        _check_compress(x, 1e-3, O.EC, out, info, t)
    for t, k in zip(xd, keep):
        t.copy_(k)
    outs2, infos2, _ = ctx.compress_batch(None, None, O.EC, prepared=prep)
    torch.cuda.synchronize()
    for x, t, out, info in zip(xs, xd, outs2, infos2):
        _check_compress(x, 1e-3, O.EC, out, info, t)


@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_error_bound_sweep_over_the_same_arrays(ctx, mode):
    """What the reference's own driver does (tests/test-dctz.sh:13-56): every file under several bounds.  Items of a batch
    that share their input share one statistics pass (calc_data_stat does not depend on the bound); every item must still
    be its own call's result -- own sf, own streams, own record -- whatever the order of the items, with scaled copies too."""
    import torch
    ctx.set_one_launch(False)                                        # (the chain of batch kernels, where the pass is shared)
    try:
        a = W.ragged(4096 * 37 + 64 * 5 + 9, np.float64, scale=37.0)
        b = W.ragged(4096 * 11 + 17, np.float32, scale=420.0)
        c = W.ragged(4096 * 5, np.float64, seed=9, scale=0.5)
        ad, bd, cd = _dev(ctx, a), _dev(ctx, b), _dev(ctx, c)
        order = [(a, ad, 1e-3), (b, bd, 1e-2), (a, ad, 1e-5), (c, cd, 1e-3), (b, bd, 1e-4), (a, ad, 1e-4), (c, cd, 1e-6), (a.copy(), ad.clone(), 1e-3)]
        xd = [t for _, t, _ in order]
        ebs = [e for _, _, e in order]
        scaled = [torch.empty_like(t) for t in xd]
        outs, infos, _ = ctx.compress_batch(xd, ebs, mode, scaled=scaled)
        torch.cuda.synchronize()
        for (x, _, eb), out, info, sc in zip(order, outs, infos, scaled):
            _check_compress(x, eb, mode, out, info, sc)
    finally:
        ctx.set_one_launch(True)


@pytest.fixture()
def spec_ctx():
    """The chain of batch kernels with speculation from 2^18 elements on (default: 2^22), so that arrays the oracle finishes
    in a moment take their scaling factor from a sample."""
    import dctz_amd
    c = dctz_amd.Context(0)
    c.set_speculation(True, 1 << 18)
    c.set_one_launch(False)
    yield c
    c.close()


def _unsampled_element(dtype, group_index, group=64):
    """An element k_stats_batch's sample does not read (one 4 KiB chunk per group of 64, at a hashed position)."""
    chunk_elems = 256 * (16 // np.dtype(dtype).itemsize)
    g0 = group_index * group + (((group_index * 2654435761) & 0xFFFFFFFF) >> 8) % group
    c = group_index * group + (0 if g0 != group_index * group else 1)
    return c * chunk_elems + 17


@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_speculative_items_of_a_batch(spec_ctx, mode):
    """Large items of a launch sequence take their scaling factor from a sample and their statistics from the compress
    kernel (as the single-array path does): the same streams, the TRUE max / min in the record; small items and an item that
    is scaled in place keep the full pass."""
    import torch
    import dctz_amd
    xs = [W.ragged(1 << 20, np.float64, scale=37.0), W.ragged((1 << 20) + 64 * 7 + 40, np.float32, scale=420.0),
          W.ragged(5000, np.float64, scale=3.0), W.ragged(300000, np.float32, seed=4, scale=0.6),
          W.ragged((1 << 19) + 33, np.float64, seed=5, scale=37.0)]
    ebs = [1e-3, 1e-4, 1e-3, 1e-3, 1e-5]
    xd = [_dev(spec_ctx, x) for x in xs]
    scaled = [torch.empty_like(xd[0]), None, None, torch.empty_like(xd[3]), xd[4]]     # the last one in place
    n0 = spec_ctx.counter(8)
    outs, infos, _ = spec_ctx.compress_batch(xd, ebs, mode, scaled=scaled)
    torch.cuda.synchronize()
    assert spec_ctx.counter(8) - n0 == 3 and spec_ctx.counter(9) == 0
    for x, eb, out, info, sc, spec in zip(xs, ebs, outs, infos, scaled, [True, True, False, True, False]):
        _check_compress(x, eb, mode, out, info, sc)
        assert bool(info.flags & dctz_amd.hip.INFO_STATS_FUSED) == spec


def test_speculative_item_with_a_wrong_guess_is_done_again(spec_ctx):
    """A spike between the sample's chunks moves max|x| into the next decade: the true statistics refuse the guess, the item
    is compressed again on its own -- the others are left as they are -- and the next batches do not speculate for a while."""
    import torch
    import dctz_amd
    xs = [W.ragged(1 << 20, np.float64, scale=37.0), W.ragged(1 << 20, np.float32, seed=2, scale=37.0), W.ragged(300000, np.float64, seed=3, scale=37.0)]
    xs[1][_unsampled_element(np.float32, 5)] = 4321.0
    xd = [_dev(spec_ctx, x) for x in xs]
    scaled = [torch.empty_like(t) for t in xd]
    outs, infos, _ = spec_ctx.compress_batch(xd, 1e-3, O.EC, scaled=scaled)
    torch.cuda.synchronize()
    assert spec_ctx.counter(8) == 3 and spec_ctx.counter(9) == 1
    cs = [_check_compress(x, 1e-3, O.EC, out, info, sc) for x, out, info, sc in zip(xs, outs, infos, scaled)]
    assert cs[1].sf == 1000.0 and cs[0].sf == 10.0
    outs, infos, _ = spec_ctx.compress_batch(xd, 1e-3, O.EC)          # the pause: full passes, same results
    torch.cuda.synchronize()
    assert spec_ctx.counter(8) == 3
    for x, out, info in zip(xs, outs, infos):
        _check_compress(x, 1e-3, O.EC, out, info)
        assert not (info.flags & dctz_amd.hip.INFO_STATS_FUSED)


def test_many_arrays_and_a_big_one(ctx):
    """More arrays of one element type than one launch sequence takes (1024; 1533 tiny fp64 ones here) and one array beyond the size from which a
    batch hands an array to the single-array path."""
    import torch
    rng = np.random.default_rng(5)
    xs = [W.ragged(int(rng.integers(1, 700)), np.float64 if i % 3 else np.float32, seed=i, scale=5.0) for i in range(2300)]
    xs.append(W.ragged((1 << 24) + 64 * 9 + 3, np.float64, scale=37.0))
    xd = [_dev(ctx, x) for x in xs]
    outs, infos, _ = ctx.compress_batch(xd, 1e-3, O.EC)
    torch.cuda.synchronize()
    import dctz_amd
    assert infos[-1].flags & dctz_amd.hip.INFO_STATS_FUSED            # the big one went down the single-array path
    cs = [_check_compress(x, 1e-3, O.EC, out, info) for x, out, info in zip(xs, outs, infos)]
    dsts, status, _ = ctx.decompress_batch(outs, [c.cnt for c in cs], [x.size for x in xs], [_tdt(x) for x in xs], 1e-3,
                                           [c.sf for c in cs], O.EC)
    torch.cuda.synchronize()
    for c, d in zip(cs, dsts):
        assert np.array_equal(_bits(d.cpu().numpy()), _bits(O.decompress(c, O.FAST)))


def test_batch_decode_reports_the_short_array(ctx):
    """An ac_count smaller than what bin_index flags (a damaged header): that array is reported, the others are
    reconstructed all the same (single call: DCTZHIP_E_ARG, dctzhip_decompress)."""
    import torch
    import dctz_amd
    xs = [np.random.default_rng(i).random(64 * 30 + 11) for i in range(4)]       # noise: plenty of exact coefficients
    xd = [_dev(ctx, x) for x in xs]
    outs, infos, _ = ctx.compress_batch(xd, 1e-3, O.EC)
    cnts = [i.cnt for i in infos]
    cnts[2] -= 1
    dsts, status, _ = ctx.decompress_batch(outs, cnts, [x.size for x in xs], [_tdt(x) for x in xs], 1e-3, [i.sf for i in infos],
                                           O.EC, check=False)
    torch.cuda.synchronize()
    assert status == [0, 0, dctz_amd.hip.E_ARG, 0]
    for j in (0, 1, 3):
        c = O.compress(xs[j], 1e-3, O.EC, O.FAST)
        assert np.array_equal(_bits(dsts[j].cpu().numpy()), _bits(O.decompress(c, O.FAST)))
    with pytest.raises(dctz_amd.hip.DctzHipError):
        ctx.decompress_batch(outs, cnts, [x.size for x in xs], [_tdt(x) for x in xs], 1e-3, [i.sf for i in infos], O.EC)
    # the context is usable afterwards
    cnts[2] += 1
    dsts, status, _ = ctx.decompress_batch(outs, cnts, [x.size for x in xs], [_tdt(x) for x in xs], 1e-3, [i.sf for i in infos], O.EC)
    assert status == [0, 0, 0, 0]


def test_c_program_drives_a_list_through_the_batch_entry_points(tmp_path):
    """tests/c/batch_list.c (the code INTEGRATION.md section E shows): built with gcc against include/dctz_hip.h alone,
    the six list-msst19 lengths x four bounds, one call per array against one batch -- byte for byte the same."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "batch_list")
    lib = os.path.join(root, "dctz_amd", "lib")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", os.path.join(root, "tests", "c", "batch_list.c"), "-I", os.path.join(root, "include"),
                           "-L", lib, "-ldctzhip", f"-Wl,-rpath,{lib}", "-lm", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "BATCH arrays=24 identical" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe, "1", "63", "64", "4097", "70001"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "BATCH arrays=20 identical" in r.stdout, r.stdout + r.stderr
