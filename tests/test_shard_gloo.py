"""N > 1 path on CPU: world_size-2 gloo run of the shard plan + stream gather
that bench.py uses with RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dctz_amd import shard
from oracle import oracle as O
from tests import workloads as W


def test_plan_shards_block_aligned():
    for total, world in ((1 << 20, 8), (1000, 3), (64 * 7 + 5, 2), (63, 4), (8 * 512 ** 3, 8)):
        plan = shard.plan_shards(total, world)
        assert len(plan) == world and sum(l for _, l in plan) == total
        off = 0
        for i, (o, l) in enumerate(plan):
            assert o == off and (o % 64 == 0 or l == 0)
            assert l % 64 == 0 or i == max(j for j, (_, ll) in enumerate(plan) if ll)   # only the last non-empty shard is ragged
            off += l
    with pytest.raises(ValueError):
        shard.plan_shards(2 ** 33, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = W.ragged(64 * 40 + 17 * rank, np.float64, seed=100 + rank, scale=37.0)   # shard per rank, ragged
        c = O.compress(x, 1e-3, O.EC, O.FAST)                                          # stand-in for the HIP stage
        streams = {"bin_index": torch.from_numpy(c.bin_index), "dc": torch.from_numpy(c.dc),
                   "ac_exact": torch.from_numpy(np.concatenate([c.ac_exact, np.zeros(5, np.float32)]))}
        got = shard.gather_streams(streams, c.cnt, dst=0)
        slow = shard.max_over_ranks(1.0 + rank, torch.device("cpu"))
        ok = slow == float(world)
        if rank == 0:
            ok = ok and len(got) == world
            for r in range(world):
                xr = W.ragged(64 * 40 + 17 * r, np.float64, seed=100 + r, scale=37.0)
                cr = O.compress(xr, 1e-3, O.EC, O.FAST)
                ok = ok and got[r]["cnt"] == cr.cnt and got[r]["n"] == xr.size
                ok = ok and np.array_equal(got[r]["bin_index"].numpy(), cr.bin_index)
                ok = ok and np.array_equal(got[r]["dc"].numpy(), cr.dc)
                ok = ok and np.array_equal(got[r]["ac_exact"].numpy(), cr.ac_exact)
                # root can decode every gathered shard independently
                cr2 = cr
                cr2.bin_index, cr2.dc, cr2.ac_exact = (got[r]["bin_index"].numpy(), got[r]["dc"].numpy(),
                                                       got[r]["ac_exact"].numpy())
                rec = O.decompress(cr2)
                ok = ok and np.abs(rec - xr).max() < 0.1
        else:
            ok = ok and got is None
        q.put((rank, bool(ok)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_gather_streams_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
