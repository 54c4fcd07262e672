#!/usr/bin/env python3
"""Multi-dimensional blocks beside the reference's flat blocks on the same device-resident array:
time per compress / decompress call (host clock around the call + a stream sync, median of the rounds), the
exception fraction p = cnt / positions, and the bytes of the three pre-zlib streams.  Not the headline metric
(bench.py): the tile mode pays a gather / scatter pass per direction in this round's implementation."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", choices=["c3_3d", "field_2d"], default="c3_3d")
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--eb", type=float, default=1e-3)
    a = ap.parse_args()
    import numpy as np
    import torch
    import dctz_amd
    from tests import workloads as W
    if a.case == "c3_3d":
        xh = W.c3(a.n).reshape(a.n, a.n, a.n)
        tdt = torch.float64
    else:                                   # CESM-like smooth 2-D fp32 field, 4 x 4 copies of the C2 stand-in's shape
        ny, nx = 1800 * 4, 3600 * 4
        y, x = np.meshgrid(np.linspace(0, 4, ny, dtype=np.float32), np.linspace(0, 4, nx, dtype=np.float32), indexing="ij")
        xh = (np.sin(6 * np.pi * x) * np.cos(4 * np.pi * y) + 0.3 * np.sin(10 * np.pi * x * y)
              + 0.01 * np.random.default_rng(2024).standard_normal((ny, nx), dtype=np.float32)).astype(np.float32)
        tdt = torch.float32
    ctx = dctz_amd.Context(0)
    xd = torch.from_numpy(xh).to(ctx.device)
    flat = xd.reshape(-1)
    n = flat.numel()
    es = flat.element_size()
    res = {}
    for name in ("flat", "tiles"):
        tc, td = [], []
        for r in range(a.rounds + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if name == "flat":
                out, info = ctx.compress(flat, a.eb, dctz_amd.EC)
            else:
                out, info = ctx.compress_nd(xd, a.eb, dctz_amd.EC)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if name == "flat":
                rec = ctx.decompress(out, info.cnt, n, tdt, a.eb, info.sf, dctz_amd.EC)
            else:
                rec = ctx.decompress_nd(out, info.cnt, xd.shape, tdt, a.eb, info.sf, dctz_amd.EC)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if r >= 2:
                tc.append((t1 - t0) * 1e3); td.append((t2 - t1) * 1e3)
        err = float((rec.reshape(-1) - flat).abs().max())
        npos = out["bin_index"].numel()
        mc, md = statistics.median(tc), statistics.median(td)
        res[name] = {"compress_ms": round(mc, 4), "decompress_ms": round(md, 4),
                     "GBps_input_roundtrip": round(n * es / ((mc + md) * 1e-3) / 1e9, 1),
                     "positions": npos, "cnt": int(info.cnt), "p": round(info.cnt / npos, 5),
                     "pre_zlib_bytes": npos + 4 * int(info.nblk) + 4 * int(info.cnt), "max_abs_err": err, "sf": info.sf}
        del out, rec
    print(json.dumps({"case": a.case, "shape": list(xh.shape), "dtype": str(xh.dtype), "eb": a.eb, "input_bytes": n * es, **res}))


if __name__ == "__main__":
    main()
