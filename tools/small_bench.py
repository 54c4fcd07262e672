#!/usr/bin/env python3
"""Latency of one compress / decompress call on small and medium inputs (the C5 lengths of tests/list-msst19.txt, C1, C2):
host clock around the call + stream sync, median of the rounds."""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import dctz_amd
    from tests import workloads as W
    ctx = dctz_amd.Context(0)
    cases = [("c5_12960_f64", W.c5_fp64(12960, 1), 1e-3), ("c5_37024_f64", W.c5_fp64(37024, 2), 1e-3),
             ("c1_1Mi_f64", W.c1(), 1e-3), ("c2_1800x3600_f32", W.c2(), 1e-4), ("c3_128_f64", W.c3(128), 1e-3)]
    for name, xh, eb in cases:
        x = torch.from_numpy(np.ascontiguousarray(xh)).to(ctx.device)
        n = x.numel()
        tdt = x.dtype
        tc, td = [], []
        for r in range(40):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out, info = ctx.compress(x, eb, dctz_amd.EC)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rec = ctx.decompress(out, info.cnt, n, tdt, eb, info.sf, dctz_amd.EC)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if r >= 8:
                tc.append((t1 - t0) * 1e6); td.append((t2 - t1) * 1e6)
        print(json.dumps({"case": name, "n": n, "bytes": n * x.element_size(), "compress_us": round(statistics.median(tc), 1),
                          "decompress_us": round(statistics.median(td), 1), "p": round(info.cnt / n, 4),
                          "GBps_roundtrip": round(n * x.element_size() / ((statistics.median(tc) + statistics.median(td)) * 1e-6) / 1e9, 2)}))


if __name__ == "__main__":
    main()
