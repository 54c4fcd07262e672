#!/usr/bin/env python3
"""Where k_dfl_parse's time goes: the entropy stage on 128 MiB of (a) zeros -- one token per segment: pass A and the fixed
costs --, (b) random bytes -- 128 literal tokens per segment: the walk at its longest --, (c) the real bin_index of the
512^3 workload.  Host clock around dctzhip_deflate of ONE section with the match search on; run under rocprofv3
--kernel-trace --stats for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dctz_amd
from tests import workloads as W

ctx = dctz_amd.Context(0)
n = 128 << 20
x = W.c3(512)
out, info = ctx.compress(torch.from_numpy(x.ravel()).to(ctx.device), 1e-3, 0)
cases = {"zeros": torch.zeros(n, dtype=torch.uint8, device=ctx.device),
         "random": torch.randint(0, 256, (n,), dtype=torch.uint8, device=ctx.device),
         "bin_index": out["bin_index"]}
for name, t in cases.items():
    if len(sys.argv) > 1 and name != sys.argv[1]:
        continue
    ctx.deflate([t], literals=[False]); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); z = ctx.deflate([t], literals=[False]); ts.append(time.perf_counter() - t0)
    print(name, "ms", round(float(np.median(ts)) * 1e3, 3), "stream bytes", int(z[0].numel()), flush=True)
