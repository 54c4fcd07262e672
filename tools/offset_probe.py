#!/usr/bin/env python3
"""k_compress / k_decompress of the 512^3 fp64 shard against the PLACEMENT of the caller's buffers: the kernels stream the
input (8 B / element), bin_index (1), DC and the lists side by side, and which HBM channels the streams hit together
depends on the buffers' base addresses.  Varies one buffer's offset inside a larger allocation at a time.
   python3 tools/offset_probe.py [--which in|bin|ac|out] [--step BYTES] [--count N]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dctz_amd
from tests import workloads as W

ap = argparse.ArgumentParser()
ap.add_argument("--which", default="in")
ap.add_argument("--step", type=int, default=4096)
ap.add_argument("--count", type=int, default=16)
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
ctx = dctz_amd.Context(0)
ctx.set_profiling(True)
x = W.c3(512, seed=512, dtype=np.float64).ravel()
n = x.size
slack = a.step * a.count + 4096
big_in = torch.empty(n * 8 + slack, dtype=torch.uint8, device=ctx.device)
big_bin = torch.empty(n + slack, dtype=torch.uint8, device=ctx.device)
big_ac = torch.empty(n * 4 + slack, dtype=torch.uint8, device=ctx.device)
big_out = torch.empty(n * 8 + slack, dtype=torch.uint8, device=ctx.device)
dc = torch.empty(n // 64, dtype=torch.float32, device=ctx.device)
xh = torch.from_numpy(x)
print(f"bases: in {big_in.data_ptr():#x} bin {big_bin.data_ptr():#x} ac {big_ac.data_ptr():#x} out {big_out.data_ptr():#x}")
for k in range(a.count):
    off = {w: 0 for w in ("in", "bin", "ac", "out")}
    off[a.which] = k * a.step
    xin = big_in[off["in"]: off["in"] + n * 8].view(torch.float64)
    xin.copy_(xh)
    out = {"bin_index": big_bin[off["bin"]: off["bin"] + n], "dc": dc, "ac_exact": big_ac[off["ac"]: off["ac"] + n * 4].view(torch.float32)}
    dst = big_out[off["out"]: off["out"] + n * 8].view(torch.float64)
    tc, td = [], []
    for r in range(a.reps):
        _, info = ctx.compress(xin, 1e-3, 0, out=out)
        tc.append(ctx.timings()["main_ms"])
        ctx.decompress(out, info.cnt, n, torch.float64, 1e-3, info.sf, 0, dst=dst)
        td.append(ctx.timings()["main_ms"])
    tc, td = np.array(tc[5:]), np.array(td[5:])
    print(f"{a.which} + {off[a.which]:8d}: k_compress {tc.mean() * 1e3:7.1f} us (min {tc.min() * 1e3:7.1f})   k_decompress {td.mean() * 1e3:7.1f} us (min {td.min() * 1e3:7.1f})", flush=True)
