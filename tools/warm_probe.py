#!/usr/bin/env python3
"""Per-step GPU time of the first steps of a fresh process (headline workload): one event behind every step, no host
synchronisation in between.  Where do the first dozens of steps lose their 5 %?  Also the kernels' own durations
(profiling on) for steps 1..5 and 60..65."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dctz_amd
from tests import workloads as W

ctx = dctz_amd.Context(0)
x = torch.from_numpy(W.c3(512)).to(ctx.device).reshape(-1)
n = x.numel()
ctx.reserve(n, torch.float64, dctz_amd.EC)
out = ctx.alloc_outputs(n)
rec = torch.empty(n, dtype=torch.float64, device=ctx.device)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 80
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
torch.cuda.synchronize()
ev[0].record()
import time
host = []
for k in range(K):
    t0 = time.perf_counter()
    _, info = ctx.compress(x, 1e-3, dctz_amd.EC, out=out)
    ctx.decompress(out, info.cnt, n, torch.float64, 1e-3, info.sf, dctz_amd.EC, dst=rec)
    host.append(time.perf_counter() - t0)
    ev[k + 1].record()
torch.cuda.synchronize()
gpu = [ev[k].elapsed_time(ev[k + 1]) for k in range(K)]
print(json.dumps({"gpu_ms_per_step": [round(g, 4) for g in gpu], "host_ms_per_step": [round(h * 1e3, 4) for h in host]}))
