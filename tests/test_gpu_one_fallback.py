"""The safety net of the one-launch kernels, fired on purpose: a launch in which a workgroup never
posts its granule -- what a workgroup that is not resident does to the others -- gives up after 20 ms, the call is run
again through the chain of kernels and comes back with the same bytes, the context stays on the chain for a while, and
the next one-launch call is clean.  dctzhip_debug_knob(ctx, 0, 1) makes workgroup 0 withhold its granule."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import workloads as W
from dctz_amd import hip as H

pytestmark = pytest.mark.gpu

ONE_CALLS, ONE_GAVE_UP, ONE_COOLDOWN = 0, 1, 2


@pytest.fixture()
def ctx():
    import dctz_amd
    c = dctz_amd.Context(0)
    yield c
    c.close()


def _dev(ctx, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)


def _same(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def _check_streams(out, info, ref, dtype, mode):
    assert info.sf == ref.sf and info.cnt == ref.cnt
    assert np.array_equal(out["bin_index"].cpu().numpy(), ref.bin_index)
    assert _same(out["dc"].cpu().numpy(), ref.dc)
    assert _same(out["ac_exact"][:ref.cnt].cpu().numpy(), ref.ac_exact)
    if mode == O.QT:
        assert _same(np.array(info.qtable[:], dtype=dtype), ref.qtable)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_compress_launch_that_gives_up_runs_through_the_chain(ctx, dtype, mode):
    import torch
    x = W.ragged(4096 * 9 + 64 * 5 + 11, dtype, scale=37.0)
    ref = O.compress(x, 1e-3, mode, O.FAST)
    xd = _dev(ctx, x)
    out, info = ctx.compress(xd, 1e-3, mode)
    assert info.flags & H.INFO_ONE_LAUNCH and ctx.counter(ONE_GAVE_UP) == 0
    _check_streams(out, info, ref, dtype, mode)
    ctx.knob(0, 1)
    out, info = ctx.compress(xd, 1e-3, mode)              # the launch gives up; the chain of kernels does the call
    torch.cuda.synchronize()
    assert not (info.flags & H.INFO_ONE_LAUNCH)
    assert ctx.counter(ONE_GAVE_UP) == 1 and ctx.counter(ONE_COOLDOWN) > 0
    assert np.array_equal(xd.cpu().numpy(), x), "input must not be modified"
    _check_streams(out, info, ref, dtype, mode)
    ctx.knob(0, 0)
    out, info = ctx.compress(xd, 1e-3, mode)              # cooling down: the chain, without a launch that could give up
    assert not (info.flags & H.INFO_ONE_LAUNCH) and ctx.counter(ONE_GAVE_UP) == 1
    _check_streams(out, info, ref, dtype, mode)
    ctx.knob(2, 0)
    out, info = ctx.compress(xd, 1e-3, mode)              # ... and one launch again, clean
    assert info.flags & H.INFO_ONE_LAUNCH and ctx.counter(ONE_GAVE_UP) == 1
    _check_streams(out, info, ref, dtype, mode)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_decompress_launch_that_gives_up_runs_through_the_chain(ctx, dtype, mode):
    import torch
    n = 4096 * 7 + 64 * 3 + 9
    x = W.ragged(n, dtype, scale=37.0)
    c = O.compress(x, 1e-3, mode, O.FAST)
    ref = O.decompress(c, O.FAST)
    out = {"bin_index": _dev(ctx, c.bin_index), "dc": _dev(ctx, c.dc), "ac_exact": _dev(ctx, c.ac_exact if c.cnt else np.zeros(4, np.float32))}
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    calls0 = ctx.counter(ONE_CALLS)
    r = ctx.decompress(out, c.cnt, n, tdt, 1e-3, c.sf, mode, qtable=c.qtable).cpu().numpy()
    assert _same(r, ref) and ctx.counter(ONE_CALLS) == calls0 + 1 and ctx.counter(ONE_GAVE_UP) == 0
    ctx.knob(0, 1)
    r = ctx.decompress(out, c.cnt, n, tdt, 1e-3, c.sf, mode, qtable=c.qtable).cpu().numpy()
    assert _same(r, ref)
    assert ctx.counter(ONE_GAVE_UP) == 1 and ctx.counter(ONE_COOLDOWN) > 0
    ctx.knob(0, 0)
    ctx.knob(2, 0)
    r = ctx.decompress(out, c.cnt, n, tdt, 1e-3, c.sf, mode, qtable=c.qtable).cpu().numpy()
    assert _same(r, ref) and ctx.counter(ONE_GAVE_UP) == 1


def test_batch_launch_that_gives_up_is_completed_by_the_chain(ctx):
    """A mixed batch (both element types, a short last block, an in-place member) whose one-launch kernels give up: every
    array's streams are the oracle's, and the member scaled in place was never in the launch that gave up."""
    import torch
    xs = [W.ragged(4096 * 3 + 17, np.float64, scale=37.0), W.ragged(4096 * 2, np.float32, scale=37.0),
          W.ragged(12960, np.float64, scale=41.0), W.ragged(4096 + 5, np.float32, scale=3.0)]
    refs = [O.compress(x, 1e-3, O.EC, O.FAST) for x in xs]
    xd = [_dev(ctx, x) for x in xs]
    scaled = [None, None, xd[2], None]                    # member 2: x / sf over the input (dctz-comp-lib.c:193-216)
    ctx.knob(0, 1)
    outs, infos, _ = ctx.compress_batch(xd, 1e-3, O.EC, scaled=scaled)
    torch.cuda.synchronize()
    assert ctx.counter(ONE_GAVE_UP) >= 1
    for i, (o, inf, ref) in enumerate(zip(outs, infos, refs)):
        _check_streams(o, inf, ref, xs[i].dtype, O.EC)
    assert _same(xd[2].cpu().numpy(), refs[2].scaled)
    assert np.array_equal(xd[0].cpu().numpy(), xs[0])


def test_in_place_call_never_takes_the_one_launch_kernel(ctx):
    import torch
    x = W.ragged(4096 * 5 + 64 + 3, np.float64, scale=37.0)
    ref = O.compress(x, 1e-3, O.EC, O.FAST)
    xd = _dev(ctx, x)
    calls0 = ctx.counter(ONE_CALLS)
    out, info = ctx.compress(xd, 1e-3, O.EC, scaled=xd)
    torch.cuda.synchronize()
    assert not (info.flags & H.INFO_ONE_LAUNCH) and ctx.counter(ONE_CALLS) == calls0
    _check_streams(out, info, ref, np.float64, O.EC)
    assert _same(xd.cpu().numpy(), ref.scaled)
