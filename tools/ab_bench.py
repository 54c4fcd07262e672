#!/usr/bin/env python3
"""A/B of kernel variants in ONE process, interleaved rounds (cdna guide rule 24).
Variants are selected through the shim's environment knobs at context creation: DCTZHIP_FASTDIV (fd),
DCTZHIP_WG_PER_CU (wg), DCTZHIP_STATS_GRID (sg).  Different BUILDS are compared by running this tool once per
library (DCTZHIP_LIBRARY=<path>) inside one gpurun call, alternating.  Prints median kernel times (HIP events)."""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--mode", default="ec")
    ap.add_argument("--eb", type=float, default=1e-3)
    ap.add_argument("--data", default="c3")
    ap.add_argument("--variants", default="fd=2;fd=1;fd=0;fd=2,wg=4")
    a = ap.parse_args()
    import numpy as np
    import torch
    import dctz_amd
    from tests import workloads as W
    npdt = np.float64 if a.dtype == "f64" else np.float32
    tdt = torch.float64 if a.dtype == "f64" else torch.float32
    mode = dctz_amd.QT if a.mode == "qt" else dctz_amd.EC
    xh = W.c3(a.n, dtype=npdt)
    if a.data == "tileconst":      # one value per tile of 4096 elements: any permutation inside a tile leaves the data as it is
        xh = np.repeat(xh.reshape(-1)[::4096], 4096).astype(npdt)
    x = torch.from_numpy(xh).cuda()
    n = x.numel()
    ctxs = []
    for spec in a.variants.split(";"):
        kv = dict(s.split("=") for s in spec.split(","))
        os.environ["DCTZHIP_FASTDIV"] = kv.get("fd", "2")
        os.environ["DCTZHIP_WG_PER_CU"] = kv.get("wg", "0")
        os.environ["DCTZHIP_STATS_GRID"] = kv.get("sg", "2048")
        os.environ["DCTZHIP_GRID_C"] = kv.get("gc", "0")
        c = dctz_amd.Context(0)
        c.set_profiling(True)
        c.reserve(n, tdt, mode)
        ctxs.append((spec, c))
    out = ctxs[0][1].alloc_outputs(n)
    rec = torch.empty(n, dtype=tdt, device="cuda")
    res = {spec: {"c": [], "d": [], "s": [], "ct": [], "dt": [], "dp": []} for spec, _ in ctxs}
    ref = None
    for r in range(a.rounds + 1):
        for spec, c in ctxs:
            _, info = c.compress(x, a.eb, mode, out=out)
            tc = c.timings()
            c.decompress(out, info.cnt, n, tdt, a.eb, info.sf, mode, qtable=np.array(info.qtable[:]), dst=rec)
            td = c.timings()
            if r == 0:   # warm-up round doubles as a cross-variant equality check
                sig = (info.cnt, int(out["bin_index"].sum(dtype=torch.int64).item()), float(rec.double().sum().item()))
                ref = ref or sig
                assert sig == ref, (spec, sig, ref)
                continue
            res[spec]["c"].append(tc["main_ms"]); res[spec]["d"].append(td["main_ms"]); res[spec]["s"].append(tc["stats_ms"])
            res[spec]["ct"].append(tc["tail_ms"]); res[spec]["dt"].append(td["tail_ms"]); res[spec]["dp"].append(td["stats_ms"])
    es = x.element_size()
    p = info.cnt / n
    bc = n * (es + 1.0625 + 4 * p)
    for spec, _ in ctxs:
        mc, md, ms = (statistics.median(res[spec][k]) for k in ("c", "d", "s"))
        print(json.dumps({"variant": spec, "compress_ms": round(mc, 4), "decompress_ms": round(md, 4), "stats_ms": round(ms, 4),
                          "compress_frac_hbm": round(bc / (mc * 1e-3) / 8e12, 4), "decompress_frac_hbm": round(bc / (md * 1e-3) / 8e12, 4),
                          "min_c": round(min(res[spec]["c"]), 4), "min_d": round(min(res[spec]["d"]), 4),
                          "compress_tail_ms": round(statistics.median(res[spec]["ct"]), 4), "decompress_count_ms": round(statistics.median(res[spec]["dp"]), 4),
                          "p": round(p, 4)}))


if __name__ == "__main__":
    main()
