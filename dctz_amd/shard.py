"""Multi-GPU plumbing for the hot path.

Shards are independent dctz_compress calls: own scaling factor, own header, no
cross-shard arithmetic.  The only exchange step is gathering the pre-zlib streams
(bin_index, DC, AC_exact) to one rank for the host zlib tail; it is done with
grouped point-to-point sends so that all inbound xGMI links of the root are used
at once (a ring all-gather would be bound by one link and move 8x the bytes).

Works with any torch.distributed backend: "nccl" (= RCCL) on GPUs, "gloo" in the
CPU tests.
"""
import torch
import torch.distributed as dist

BLK = 64
MAX_ELEMS = 2 ** 31 - 1          # dctz.h:126: N is an int; header counts are uint32


def plan_shards(total_elems, world_size):
    """Split one flat array into world_size contiguous shards on 64-element
    boundaries (a block never straddles two shards); returns [(offset, length)].
    The last shard takes the remainder block.  Every shard must fit an int."""
    nblk = (total_elems + BLK - 1) // BLK
    base, extra = divmod(nblk, world_size)
    plan, off = [], 0
    for r in range(world_size):
        blocks = base + (1 if r < extra else 0)
        length = min(blocks * BLK, total_elems - off)
        if length > MAX_ELEMS:
            raise ValueError("shard exceeds 2^31-1 elements; use more shards")
        plan.append((off, length))
        off += length
    assert off == total_elems
    return plan


def max_over_ranks(value, device, group=None):
    """Slowest rank's time (the bench contract's MAX over ranks)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_streams(streams, cnt, dst=0, group=None):
    """Send this rank's pre-zlib streams to rank `dst`.

    streams: {"bin_index": uint8[n], "dc": float32[nblk], "ac_exact": float32[>=cnt]}
    Returns, on dst, a list (one entry per rank, own entry = the input tensors) of
    {"bin_index", "dc", "ac_exact"(trimmed to that rank's cnt), "cnt", "n"};
    None elsewhere."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = streams["bin_index"].device
    meta = torch.tensor([streams["bin_index"].numel(), streams["dc"].numel(), int(cnt)],
                        dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    ac = streams["ac_exact"][:int(cnt)]
    ops, result = [], None
    if rank == dst:
        result = []
        for r in range(world):
            n_r, nblk_r, cnt_r = (int(v) for v in metas[r].tolist())
            if r == rank:
                result.append({"bin_index": streams["bin_index"], "dc": streams["dc"], "ac_exact": ac,
                               "cnt": cnt_r, "n": n_r})
                continue
            e = {"bin_index": torch.empty(n_r, dtype=torch.uint8, device=dev),
                 "dc": torch.empty(nblk_r, dtype=torch.float32, device=dev),
                 "ac_exact": torch.empty(cnt_r, dtype=torch.float32, device=dev), "cnt": cnt_r, "n": n_r}
            result.append(e)
            ops += [dist.P2POp(dist.irecv, e["bin_index"], r, group), dist.P2POp(dist.irecv, e["dc"], r, group)]
            if cnt_r:
                ops.append(dist.P2POp(dist.irecv, e["ac_exact"], r, group))
    else:
        ops += [dist.P2POp(dist.isend, streams["bin_index"], dst, group),
                dist.P2POp(dist.isend, streams["dc"], dst, group)]
        if int(cnt):
            ops.append(dist.P2POp(dist.isend, ac, dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return result
