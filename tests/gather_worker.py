"""One rank of tests/test_gpu_gather_double.py: compresses its own shard on cuda:0, joins the communicator (the RCCL test
double of tests/c/rccl_double.cpp: several processes of ONE GPU), takes part in dctzhip_comm_sizes + dctzhip_comm_gather,
and writes what it saw to <dir>/rank<r>.npz.  argv: rank world root dir [mode]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, root, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import dctz_amd
    from tests import workloads as W
    ctx = dctz_amd.Context(0)
    # shards of different lengths (one with a remainder block, one without exceptions to speak of), like C4's seeds
    n = [64 * 700 + 9, 64 * 1300, 64 * 257 + 40, 64 * 64, 64 * 999 + 1, 64 * 3][rank]
    dtype = np.float64
    x = W.ragged(n, dtype, scale=37.0 + rank) if rank != 3 else np.full(n, 2.5, dtype)
    out, info = ctx.compress(torch.from_numpy(x).to(ctx.device), 1e-3, dctz_amd.EC)
    idf = os.path.join(d, "id.bin")
    if rank == 0:
        uid = dctz_amd.Context.comm_unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.monotonic()
        while not os.path.exists(idf):
            if time.monotonic() - t0 > 60:
                raise SystemExit("no id file")
            time.sleep(0.01)
        uid = open(idf, "rb").read()
    ctx.comm_create(rank, world, uid)
    rc, err, got = 0, "", None
    try:
        for _ in range(2):                                    # twice: the channels and the kept receive buffers are reused
            got = ctx.comm_gather(out, info.cnt, n, root=root)
    except dctz_amd.DctzHipError as e:
        rc, err = 1, str(e)
    finally:
        ctx.lib.dctzhip_comm_destroy(ctx.h)
    rec = {"n": n, "cnt": info.cnt, "rc": rc, "err": err,
           "bin": out["bin_index"].cpu().numpy(), "dc": out["dc"].cpu().numpy(), "ac": out["ac_exact"][:info.cnt].cpu().numpy()}
    if got is not None:
        rec.update(all_bin=got["bin_index"].cpu().numpy(), all_dc=got["dc"].cpu().numpy(), all_ac=got["ac_exact"].cpu().numpy(),
                   sizes=np.array(got["sizes"], dtype=np.uint64))
    np.savez(os.path.join(d, f"rank{rank}.npz"), **rec)
    ctx.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
