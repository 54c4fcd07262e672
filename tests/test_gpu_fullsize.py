"""BASELINE.json full-size configuration (C3/C4 shard: 512^3 fp64 = 1 GiB) on the GPU.
The oracle takes ~20 s on this size, so parity here is by size-independent
properties plus one oracle comparison of the streams' digests:
  * stream consistency: #(bin == 255) = cnt + nblk, block heads are 255;
  * the two independent exception-placement schemes (two-level vs single-pass
    look-back) give byte-identical streams and reconstructions;
  * round trip honours the error bound;
  * EC and QT agree on everything that does not depend on the table;
  * (slow, still bounded) the HIP streams equal the oracle's on the full shard."""
import hashlib
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


def _ctx(feat):
    import dctz_amd
    os.environ["DCTZHIP_FEAT"] = str(feat)
    try:
        return dctz_amd.Context(0)
    finally:
        os.environ.pop("DCTZHIP_FEAT", None)


def _digest(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()


def test_full_shard_properties_and_scheme_agreement(shard):
    import torch
    x = torch.from_numpy(shard).cuda()
    n = x.numel()
    nblk = n // 64
    eb = 1e-3
    res = {}
    for feat in (0, 1):
        ctx = _ctx(feat)
        out, info = ctx.compress(x, eb, 0)
        rec = ctx.decompress(out, info.cnt, n, torch.float64, eb, info.sf, 0)
        res[feat] = (info.cnt, info.sf, _digest(out["bin_index"]), _digest(out["dc"]),
                     _digest(out["ac_exact"][:info.cnt]), _digest(rec))
        if feat == 0:
            b = out["bin_index"]
            assert int((b == 255).sum().item()) == info.cnt + nblk
            assert bool((b[::64] == 255).all().item())
            err = (rec - (x / info.sf) * info.sf).abs().max().item()
            assert err <= 8.5 * eb * info.sf
            # QT on the same data: same bins / DC / count, table covers every flagged coefficient
            outq, infoq = ctx.compress(x, eb, 1)
            assert infoq.cnt == info.cnt and _digest(outq["bin_index"]) == res[0][2] and _digest(outq["dc"]) == res[0][3]
            q = np.array(infoq.qtable[:])
            assert np.all(q[1:] >= 1.0) and q[1:].max() <= 80.0      # |coef| <= sqrt(64) * max|x/sf| <= 80
            recq = ctx.decompress(outq, infoq.cnt, n, torch.float64, eb, infoq.sf, 1, qtable=q)
            assert (recq - (x / info.sf) * info.sf).abs().max().item() <= 8.5 * eb * info.sf
        ctx.close()
    assert res[0] == res[1], "two-level and single-pass schemes must produce identical bytes"


def test_full_shard_matches_oracle_digests(shard):
    """~25 s of CPU: the whole 1 GiB shard through the oracle, compared by digest."""
    import torch
    ctx = _ctx(0)
    x = torch.from_numpy(shard).cuda()
    out, info = ctx.compress(x, 1e-3, 0)
    c = O.compress(shard, 1e-3, O.EC, O.FAST)
    assert (info.cnt, info.sf) == (c.cnt, c.sf)
    assert _digest(out["bin_index"]) == hashlib.sha256(c.bin_index.tobytes()).hexdigest()
    assert _digest(out["dc"]) == hashlib.sha256(c.dc.tobytes()).hexdigest()
    assert _digest(out["ac_exact"][:c.cnt]) == hashlib.sha256(c.ac_exact.tobytes()).hexdigest()
    rec = ctx.decompress(out, info.cnt, x.numel(), torch.float64, 1e-3, info.sf, 0)
    assert _digest(rec) == hashlib.sha256(O.decompress(c, O.FAST).tobytes()).hexdigest()
    ctx.close()
