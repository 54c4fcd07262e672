#!/usr/bin/env python3
"""Phase timers of k_compress<double> (diagnostic builds only: make -C dctz_amd hip LIBDIR=... EXTRA=-DDCTZ_STAMP, then
DCTZHIP_LIBRARY=<that build> python3 tools/stamp_report.py).  The kernel adds up s_memtime differences of thread 0 of
every workgroup per phase; printed: cycles per tile and wave (two waves share a SIMD, so every figure includes the
partner's share of the issue slots)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["regs of the first half (xn -> x)", "wait: second half landed", "LDS -> registers (second half)",
         "issue DMA: first half of the next tile", "flush of the previous tile", "statistics + scaling (second half)",
         "transform", "binning, coefficients 0-31", "wait: first half of the next tile landed",
         "LDS -> registers + issue DMA (second half of the next tile) + statistics / scaling", "binning, coefficients 32-63",
         "(after the loop) last flush"]


def main():
    import numpy as np
    import torch
    import dctz_amd
    from tests import workloads as W
    n_edge = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ctx = dctz_amd.Context(0)
    lib = ctx.lib
    lib.dctzhip_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    x = torch.from_numpy(W.c3(n_edge)).to(ctx.device)
    n = x.numel()
    out = ctx.alloc_outputs(n)
    for _ in range(3):
        ctx.compress(x, 1e-3, dctz_amd.EC, out=out)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 12)()
    lib.dctzhip_debug_stamps(ctx.h, buf)                 # reset
    reps = 10
    for _ in range(reps):
        ctx.compress(x, 1e-3, dctz_amd.EC, out=out)
    torch.cuda.synchronize()
    lib.dctzhip_debug_stamps(ctx.h, buf)
    tiles = (n // 4096) * reps
    tot = 0.0
    for i in range(12):
        v = buf[i] / tiles
        tot += v
        print(f"{i:2d}  {v:9.0f} cycles/tile  {NAMES[i]}")
    print(f"    {tot:9.0f} cycles per tile and wave in all (s_memtime ticks)")
    # ---- k_decompress<double>: one wave per SIMD ----
    dn = ["loop head", "wait: bin ids / DC / exact coefficients of this tile landed", "flag counts + wave scan",
          "de-quantisation (bin centres, exact coefficients)", "prefetch of the next tile (issue)", "inverse transform",
          "de-scaling", "registers -> LDS image -> 32 row stores"]
    if hasattr(lib, "dctzhip_debug_stamps_dec"):
        lib.dctzhip_debug_stamps_dec.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
        _, info = ctx.compress(x, 1e-3, dctz_amd.EC, out=out)
        rec = torch.empty(n, dtype=torch.float64, device=ctx.device)
        for _ in range(3):
            ctx.decompress(out, info.cnt, n, torch.float64, 1e-3, info.sf, dctz_amd.EC, dst=rec)
        lib.dctzhip_debug_stamps_dec(ctx.h, buf)
        for _ in range(reps):
            ctx.decompress(out, info.cnt, n, torch.float64, 1e-3, info.sf, dctz_amd.EC, dst=rec)
        lib.dctzhip_debug_stamps_dec(ctx.h, buf)
        tot = 0.0
        print("k_decompress<double>:")
        for i in range(8):
            v = buf[i] / tiles
            tot += v
            print(f"{i:2d}  {v:9.0f} cycles/tile  {dn[i]}")
        print(f"    {tot:9.0f} cycles per tile and wave in all")


if __name__ == "__main__":
    main()
