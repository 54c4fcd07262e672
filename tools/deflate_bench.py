#!/usr/bin/env python3
"""Timing of the GPU entropy stage on the streams of one compress call (device resident), against zlib on the host.
  python tools/deflate_bench.py [--n 512] [--eb 1e-3] [--mode ec|qt] [--reps 10]"""
import argparse, json, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dctz_amd
from tests import workloads as W

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--eb", type=float, default=1e-3)
ap.add_argument("--mode", default="ec")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--host-sample", type=int, default=32 << 20, help="bytes of bin_index timed through zlib on one host core")
a = ap.parse_args()
ctx = dctz_amd.Context(0)
x = W.c3(a.n, dtype=np.float64 if a.dtype == "f64" else np.float32)
xd = torch.from_numpy(x.ravel()).to(ctx.device)
out, info = ctx.compress(xd, a.eb, 1 if a.mode == "qt" else 0)
cnt = int(info.cnt)
secs = [out["bin_index"], out["dc"], out["ac_exact"][:cnt]]
raw = [t.numel() * t.element_size() for t in secs]
LIT = [False, True, True]      # DC and AC_exact are bytes of floats: no match search (what the drop-in passes)
zs = ctx.deflate(secs, literals=LIT)
torch.cuda.synchronize()
ts = []
for _ in range(a.reps):
    t0 = time.perf_counter()
    zs = ctx.deflate(secs, literals=LIT)
    ts.append(time.perf_counter() - t0)
# the reader's side: the same sections inflated on the device, each on its own and all three in one call
zs, index = ctx.deflate(secs, want_index=True, literals=LIT)
inf = {}
for name, sel in (("bin_index", [0]), ("dc", [1]), ("ac_exact", [2]), ("all", [0, 1, 2])):
    tt = []
    for _ in range(max(3, a.reps // 2)):
        t0 = time.perf_counter()
        outs, ok = ctx.inflate([zs[i] for i in sel], [index[i] for i in sel], [raw[i] for i in sel])
        tt.append(time.perf_counter() - t0)
        assert ok
    inf[name] = float(np.median(tt) * 1e3)
for o, t in zip(outs, secs):
    assert torch.equal(o, t.view(torch.uint8).reshape(-1)[:o.numel()])
host = [t.cpu().numpy().tobytes() for t in secs]
for z, h in zip(zs, host):
    assert zlib.decompress(z.cpu().numpy().tobytes()) == h
sample = host[0][:a.host_sample]
t0 = time.perf_counter(); zl = len(zlib.compress(sample, 6)); t_host = time.perf_counter() - t0
ours_sample = None
res = {"workload": f"c3 {a.n}^3 {a.dtype} {a.mode} eb {a.eb}", "raw_bytes": raw, "gpu_stream_bytes": [int(z.numel()) for z in zs],
       "gpu_ms_median": float(np.median(ts) * 1e3), "gpu_ms_min": float(min(ts) * 1e3),
       "gpu_GBps_of_streams": sum(raw) / np.median(ts) / 1e9,
       "input_GBps": x.nbytes / np.median(ts) / 1e9,
       "gpu_inflate_ms": inf,
       "host_zlib6_one_core_MBps": len(sample) / t_host / 1e6, "host_sample_bytes": len(sample),
       "host_zlib6_sample_ratio": len(sample) / zl}
print(json.dumps(res))
