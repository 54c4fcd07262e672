"""BASELINE.json full-size configuration (C3/C4 shard: 512^3 fp64 = 1 GiB) on the GPU.
The oracle takes ~20 s on this size, so parity here is by size-independent
properties plus one oracle comparison of the streams' digests:
  * stream consistency: #(bin == 255) = cnt + nblk, block heads are 255;
  * round trip honours the error bound;
  * EC and QT agree on everything that does not depend on the table;
  * (slow, still bounded) the HIP streams equal the oracle's on the full shard."""
import hashlib
import dctz_amd
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import workloads as W

pytestmark = pytest.mark.gpu
N_EDGE = int(os.environ.get("DCTZ_FULLSIZE_EDGE", "512"))


@pytest.fixture(scope="module")
def shard():
    return W.c3(N_EDGE)


def _ctx(_unused=0):
    return dctz_amd.Context(0)


def _digest(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()


def test_full_shard_properties(shard):
    import torch
    x = torch.from_numpy(shard).cuda()
    n = x.numel()
    nblk = n // 64
    eb = 1e-3
    ctx = _ctx()
    out, info = ctx.compress(x, eb, 0)
    rec = ctx.decompress(out, info.cnt, n, torch.float64, eb, info.sf, 0)
    b = out["bin_index"]
    assert int((b == 255).sum().item()) == info.cnt + nblk
    assert bool((b[::64] == 255).all().item())
    err = (rec - (x / info.sf) * info.sf).abs().max().item()
    assert err <= 8.5 * eb * info.sf
    # QT on the same data: same bins / DC / count, table covers every flagged coefficient
    outq, infoq = ctx.compress(x, eb, 1)
    assert infoq.cnt == info.cnt and _digest(outq["bin_index"]) == _digest(out["bin_index"]) and _digest(outq["dc"]) == _digest(out["dc"])
    q = np.array(infoq.qtable[:])
    assert np.all(q[1:] >= 1.0) and q[1:].max() <= 80.0      # |coef| <= sqrt(64) * max|x/sf| <= 80
    recq = ctx.decompress(outq, infoq.cnt, n, torch.float64, eb, infoq.sf, 1, qtable=q)
    assert (recq - (x / info.sf) * info.sf).abs().max().item() <= 8.5 * eb * info.sf
    ctx.close()


@pytest.mark.parametrize("mode", [O.EC, O.QT], ids=["C4_shard_EC", "C3_QT"])
def test_full_shard_matches_oracle_digests(shard, mode):
    """~25 s of CPU per mode: the whole 1 GiB volume through the oracle, compared by digest -- EC (the C4 shard of
    BASELINE.json's metric) and QT (config C3)."""
    import torch
    ctx = _ctx(0)
    x = torch.from_numpy(shard).cuda()
    out, info = ctx.compress(x, 1e-3, mode)
    c = O.compress(shard, 1e-3, mode, O.FAST)
    assert (info.cnt, info.sf) == (c.cnt, c.sf)
    assert _digest(out["bin_index"]) == hashlib.sha256(c.bin_index.tobytes()).hexdigest()
    assert _digest(out["dc"]) == hashlib.sha256(c.dc.tobytes()).hexdigest()
    assert _digest(out["ac_exact"][:c.cnt]) == hashlib.sha256(c.ac_exact.tobytes()).hexdigest()
    if mode == O.QT:
        assert np.array_equal(np.array(info.qtable[:]).view(np.uint8), c.qtable.view(np.uint8))
    rec = ctx.decompress(out, info.cnt, x.numel(), torch.float64, 1e-3, info.sf, mode, qtable=np.array(info.qtable[:]))
    assert _digest(rec) == hashlib.sha256(O.decompress(c, O.FAST).tobytes()).hexdigest()
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_maximum_size_array_is_periodic(dtype):
    """Largest legal input: N is an int (dctz.h:126), so N <= 2^31 - 1 (16 GiB of doubles).
    The oracle cannot run that, but a PERIODIC input must give periodic streams: the array is a
    pattern of 64*16383 elements (whole blocks, deliberately not a multiple of the 1024-element
    tile) repeated 2048 times plus a ragged tail with a short last block, and a small array
    [pattern | tail] -- checked against the oracle -- predicts every period of the big one.
    Exercises 64-bit addressing (> 2^32 bytes), the u32 exception counters near their top and
    the remainder block at the far end."""
    import torch
    import dctz_amd
    ctx = dctz_amd.Context(0)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    eb = 1e-3
    chunk = 64 * 16383
    n = 2**31 - 9                                        # N % 64 = 55: a short last block at the very end
    reps, tail = divmod(n, chunk)
    assert tail % 64 == 55 and reps == 2048
    pat = W.ragged(chunk, dtype, scale=37.0)
    small = np.concatenate([pat, pat[:tail]])
    c = O.compress(small, eb, O.EC, O.FAST)               # the oracle on [pattern | tail]
    ref_small = O.decompress(c, O.FAST)
    out_s, info_s = ctx.compress(torch.from_numpy(small).cuda(), eb, O.EC)
    assert info_s.cnt == c.cnt and info_s.sf == c.sf
    assert np.array_equal(out_s["bin_index"].cpu().numpy(), c.bin_index)
    # exceptions of the pattern period and of the tail
    nb_pat = chunk // 64
    flags = c.bin_index == 255
    flags[::64] = False
    cnt_pat = int(flags[:chunk].sum())
    cnt_tail = c.cnt - cnt_pat

    p_dev = torch.from_numpy(pat).cuda()
    x = torch.empty(n, dtype=tdt, device="cuda")
    x[:reps * chunk].view(reps, chunk).copy_(p_dev.unsqueeze(0).expand(reps, chunk))
    x[reps * chunk:] = p_dev[:tail]
    out, info = ctx.compress(x, eb, O.EC)
    assert info.sf == c.sf
    assert info.cnt == reps * cnt_pat + cnt_tail
    assert info.cnt > 2**27                               # a few 10^8 exceptions: counters well exercised
    b = out["bin_index"]
    assert bool((b[:reps * chunk].view(reps, chunk) == out_s["bin_index"][:chunk].unsqueeze(0)).all().item())
    assert torch.equal(b[reps * chunk:], out_s["bin_index"][chunk:])
    d = out["dc"]
    assert torch.equal(d[:reps * nb_pat].view(reps, nb_pat).view(torch.int32),
                       out_s["dc"][:nb_pat].view(torch.int32).unsqueeze(0).expand(reps, nb_pat))
    assert torch.equal(d[reps * nb_pat:].view(torch.int32), out_s["dc"][nb_pat:].view(torch.int32))
    a = out["ac_exact"][:info.cnt]
    a_s = out_s["ac_exact"][:c.cnt]
    assert torch.equal(a[:reps * cnt_pat].view(reps, cnt_pat).view(torch.int32),
                       a_s[:cnt_pat].view(torch.int32).unsqueeze(0).expand(reps, cnt_pat))
    assert torch.equal(a[reps * cnt_pat:].view(torch.int32), a_s[cnt_pat:].view(torch.int32))

    del x
    rec = ctx.decompress(out, info.cnt, n, tdt, eb, info.sf, O.EC)
    r_s = torch.from_numpy(ref_small).cuda()
    it = torch.int64 if dtype == np.float64 else torch.int32
    assert torch.equal(rec[:reps * chunk].view(reps, chunk).view(it), r_s[:chunk].view(it).unsqueeze(0).expand(reps, chunk))
    assert torch.equal(rec[reps * chunk:].view(it), r_s[chunk:].view(it))
    ctx.close()


def test_entropy_stage_on_a_section_beyond_2_gib():
    """Section offsets are 64-bit throughout: 2^31 + 12345 bytes (a periodic skewed pattern made on the device) through
    dctzhip_deflate; checked by the device decoder (every chunk, lengths, adler32 of the content), by the size-independent
    structure of the stream (frame, index that tiles it) and by zlib on the first and last chunks."""
    import torch
    import zlib
    ctx = _ctx()
    n = (1 << 31) + 12345
    chunk = 16384
    period = 1 << 20                                         # 1 MiB of skewed bytes, repeated (the chunks never see the repeat: it is 64 chunks away)
    rng = np.random.default_rng(8)
    base = torch.from_numpy(rng.choice([127, 128, 126, 255, 0, 9, 200], p=[.55, .2, .1, .05, .04, .03, .03], size=period).astype(np.uint8)).to(ctx.device)
    src = base.repeat((n + period - 1) // period)[:n].contiguous()
    src[-5000:] = 77                                         # the tail differs from the pattern
    (z,), (ix,) = ctx.deflate([src], want_index=True)
    nch = (n + chunk - 1) // chunk
    assert len(ix) == nch and int(ix.astype(np.uint64).sum()) + 8 == z.numel()
    head = z[:2].cpu().numpy().tobytes(); tail = z[-6:].cpu().numpy().tobytes()
    assert head == b"\x78\x5e" and tail[:2] == b"\x03\x00"
    (back,), ok = ctx.inflate([z], [ix], [n])
    assert ok and torch.equal(back, src)
    offs = np.concatenate([[0], np.cumsum(ix.astype(np.uint64))]).astype(np.int64)
    for c in (0, 1, nch // 2, nch - 2, nch - 1):             # zlib on single chunks (raw deflate, byte aligned)
        d = zlib.decompressobj(-15)
        piece = d.decompress(z[2 + offs[c]:2 + offs[c + 1]].cpu().numpy().tobytes())
        want = src[c * chunk:min(n, (c + 1) * chunk)].cpu().numpy().tobytes()
        assert piece == want and d.unused_data == b""
    ctx.close()
