#!/usr/bin/env python3
"""Latency of the calls on small and medium inputs (the C5 lengths of tests/list-msst19.txt, C1, C2): host clock around the
call + stream sync, median of the rounds -- one call per array, and the same arrays as ONE batch call
(dctzhip_compress_batch / dctzhip_decompress_batch).

  python3 tools/small_bench.py                 every case, JSON lines
  python3 tools/small_bench.py --only batch25  just the 25-array batch, many rounds (the command the rocprofv3 kernel
                                               trace of profiles/r03_small_calls_kernel_stats.csv was taken on)
"""
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
    ap.add_argument("--only", default="")
    ap.add_argument("--rounds", type=int, default=40)
    a = ap.parse_args()
    import numpy as np
    import torch
    import dctz_amd
    from tests import workloads as W
    ctx = dctz_amd.Context(0)

    def med(v):
        return round(statistics.median(v), 1)

    def single(name, xh, eb):
        x = torch.from_numpy(np.ascontiguousarray(xh)).to(ctx.device)
        n, tdt = x.numel(), x.dtype
        tc, td = [], []
        for r in range(a.rounds):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out, info = ctx.compress(x, eb, dctz_amd.EC)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ctx.decompress(out, info.cnt, n, tdt, eb, info.sf, dctz_amd.EC)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if r >= 8:
                tc.append((t1 - t0) * 1e6); td.append((t2 - t1) * 1e6)
        print(json.dumps({"case": name, "n": n, "bytes": n * x.element_size(), "compress_us": med(tc), "decompress_us": med(td),
                          "p": round(info.cnt / n, 4),
                          "GBps_roundtrip": round(n * x.element_size() / ((statistics.median(tc) + statistics.median(td)) * 1e-6) / 1e9, 2)}), flush=True)

    def batch(name, xhs, ebs, mode=dctz_amd.EC):
        xs = [torch.from_numpy(np.ascontiguousarray(x)).to(ctx.device) for x in xhs]
        ns, tdt = [x.numel() for x in xs], [x.dtype for x in xs]
        outs, infos, cp = ctx.compress_batch(xs, ebs, mode)
        _, _, dp = ctx.decompress_batch(outs, [i.cnt for i in infos], ns, tdt, ebs, [i.sf for i in infos], mode,
                                        qtables=[np.array(i.qtable[:]) for i in infos] if mode == dctz_amd.QT else None)
        tc, td, tl = [], [], []
        for r in range(a.rounds):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.compress_batch(None, None, mode, prepared=cp)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ctx.decompress_batch(None, None, None, None, None, None, mode, prepared=dp)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if r >= 8:
                tc.append((t1 - t0) * 1e6); td.append((t2 - t1) * 1e6)
        for r in range(max(6, a.rounds // 8)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j, x in enumerate(xs):
                o, ij = ctx.compress(x, ebs[j], mode, out=outs[j])
                ctx.decompress(o, ij.cnt, ns[j], tdt[j], ebs[j], ij.sf, mode, qtable=ij.qtable if mode == dctz_amd.QT else None)
            torch.cuda.synchronize()
            if r >= 2:
                tl.append((time.perf_counter() - t0) * 1e6)
        by = sum(x.numel() * x.element_size() for x in xs)
        b_us = statistics.median(tc) + statistics.median(td)
        print(json.dumps({"case": name, "arrays": len(xs), "bytes": by, "compress_batch_us": med(tc), "decompress_batch_us": med(td),
                          "looped_calls_us": med(tl), "speedup_of_the_batch": round(statistics.median(tl) / b_us, 2),
                          "GBps_roundtrip_batch": round(by / (b_us * 1e-6) / 1e9, 2),
                          "GBps_roundtrip_looped": round(by / (statistics.median(tl) * 1e-6) / 1e9, 2)}), flush=True)

    msst = [(W.c5_fp64(n, 100 + i), eb) for i, n in enumerate(W.MSST19_LENGTHS) for eb in (1e-3, 1e-4, 1e-5, 1e-6)]
    cases = {
        "c5_12960_f64": lambda: single("c5_12960_f64", W.c5_fp64(12960, 1), 1e-3),
        "c5_37024_f64": lambda: single("c5_37024_f64", W.c5_fp64(37024, 2), 1e-3),
        "c1_1Mi_f64": lambda: single("c1_1Mi_f64", W.c1(), 1e-3),
        "c2_1800x3600_f32": lambda: single("c2_1800x3600_f32", W.c2(), 1e-4),
        "c3_128_f64": lambda: single("c3_128_f64", W.c3(128), 1e-3),
        "batch24": lambda: batch("batch24: list-msst19 lengths x eb 1e-3..1e-6 (fp64)", [m[0] for m in msst], [m[1] for m in msst]),
        "batch25": lambda: batch("batch25: batch24 + the C2 field (fp32, eb 1e-4)", [m[0] for m in msst] + [W.c2()], [m[1] for m in msst] + [1e-4]),
        "batch25_qt": lambda: batch("batch25, QT mode (what tests/test-dctz.sh runs)", [m[0] for m in msst] + [W.c2()], [m[1] for m in msst] + [1e-4], dctz_amd.QT),
    }
    for nm, fn in cases.items():
        if a.only and nm != a.only:
            continue
        fn()


if __name__ == "__main__":
    main()
