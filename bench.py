#!/usr/bin/env python3
"""bench.py -- DCTZ hot path on MI355X: compress + decompress throughput.

One "step" = one dctzhip_compress() followed by one dctzhip_decompress() of one
device-resident shard (the block-DCT + binning path of dctz_compress /
dctz_decompress; zlib tail excluded -- it stays on the host).

Workload (BASELINE.json metric "fp64 1e-3 EC"): one C4 shard per GPU = synthetic
fp64 512^3 volume (C3 formula, seed 512+rank), EC mode, error bound 1e-3.
Weak scaling: every rank owns its shard, no data-path collective (shards are
independent dctz_compress calls).  `--gather` additionally times the one real
exchange step (RCCL send/recv of the pre-zlib streams to rank 0), outside the
timed steps.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=512, help="volume edge (512 -> 1 GiB fp64 shard)")
    ap.add_argument("--eb", type=float, default=1e-3)
    ap.add_argument("--mode", choices=["ec", "qt"], default="ec")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--gather", action="store_true", help="also time the RCCL gather of the streams to rank 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-speculation", action="store_true", help="always run the separate statistics pass first")
    ap.add_argument("--cpu-sample", type=int, default=1 << 27, help="elements of the shard the CPU oracle is timed on")
    ap.add_argument("--cpu-repeats", type=int, default=4, help="passes of the CPU oracle over the sample (about 10 s in all)")
    return ap.parse_args()


def main():
    a = parse()
    import numpy as np
    import torch
    import dctz_amd
    from tests import workloads as W

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        a.gpus = world
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = local_rank if world > 1 else 0
    ctx = dctz_amd.Context(dev)

    np_dtype = np.float64 if a.dtype == "f64" else np.float32
    t_dtype = torch.float64 if a.dtype == "f64" else torch.float32
    mode = dctz_amd.QT if a.mode == "qt" else dctz_amd.EC
    x_host = W.c3(a.n, seed=512 + rank, dtype=np_dtype)
    n = x_host.size
    es = x_host.itemsize
    x = torch.from_numpy(x_host).to(ctx.device)
    out = ctx.alloc_outputs(n)
    rec = torch.empty(n, dtype=t_dtype, device=ctx.device)
    ctx.reserve(n, t_dtype, mode)
    if a.no_speculation:
        ctx.set_speculation(False)

    qt = mode == dctz_amd.QT

    def step():
        _, info = ctx.compress(x, a.eb, mode, out=out)
        ctx.decompress(out, info.cnt, n, t_dtype, a.eb, info.sf, mode, qtable=info.qtable if qt else None, dst=rec)
        return info

    info = None
    for _ in range(a.warmup):
        info = step()
    if info is None:
        info = step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps, profiling off (no event overhead) ------
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        info = step()
    barrier()
    elapsed = time.perf_counter() - t0
    from dctz_amd import shard
    elapsed = shard.max_over_ranks(elapsed, ctx.device)
    ms_per_step = elapsed * 1e3 / a.steps

    # ---- per-kernel durations (HIP events on the launch stream), same K steps --
    ctx.set_profiling(True)
    acc = {"c_stats": 0.0, "c_main": 0.0, "c_tail": 0.0, "d_main": 0.0, "d_tail": 0.0}
    t_c = t_d = 0.0
    for _ in range(a.steps):
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        _, info = ctx.compress(x, a.eb, mode, out=out)
        s1 = time.perf_counter()
        tm = ctx.timings()
        acc["c_stats"] += tm["stats_ms"]; acc["c_main"] += tm["main_ms"]; acc["c_tail"] += tm["tail_ms"]
        s2 = time.perf_counter()
        ctx.decompress(out, info.cnt, n, t_dtype, a.eb, info.sf, mode, qtable=info.qtable if qt else None, dst=rec)
        s3 = time.perf_counter()
        tm = ctx.timings()
        acc["d_main"] += tm["main_ms"]; acc["d_tail"] += tm["tail_ms"]
        t_c += s1 - s0; t_d += s3 - s2
    ctx.set_profiling(False)
    for k in acc:
        acc[k] /= a.steps
    t_c = t_c * 1e3 / a.steps
    t_d = t_d * 1e3 / a.steps

    fused = bool(info.flags & dctz_amd.hip.INFO_STATS_FUSED)     # statistics computed inside k_compress (guess verified)
    p = info.cnt / n                                            # exception fraction
    # algorithmic bytes per element: transform pass of compress reads s,
    # writes 1 (bin) + 4/64 (DC) + 4p (AC_exact); decompress reads 1 + 4/64 + 4p, writes s
    bytes_c_main = n * (es + 1.0 + 4.0 / 64.0 + 4.0 * p)
    bytes_stats = n * es
    bytes_d_main = n * (es + 1.0 + 4.0 / 64.0 + 4.0 * p)
    ach_c = bytes_c_main / (acc["c_main"] * 1e-3) / 1e9
    ach_d = bytes_d_main / (acc["d_main"] * 1e-3) / 1e9
    ach_s = bytes_stats / (acc["c_stats"] * 1e-3) / 1e9

    # HBM traffic of the dominant kernel from the PMC pass of the SAME command
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 correction
    # applied: profiles/r01_pmc_traffic.json says how); null if no matching record
    traffic = None
    try:
        rec_t = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        key = f"{a.dtype}_{a.n}_{a.mode}_{a.eb:g}" + ("_fused" if fused else "")
        if key in rec_t.get("k_compress", {}):
            traffic = rec_t["k_compress"][key]["hbm_bytes_per_launch"]
    except (OSError, ValueError):
        pass

    # ---- optional: the one real exchange step (streams -> rank 0 over RCCL) ----
    gather_ms = None
    if a.gather and dist is not None:
        from dctz_amd import shard
        barrier()
        g0 = time.perf_counter()
        shard.gather_streams(out, info.cnt, dst=0)
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3

    # ---- CPU baseline: the oracle (a port), one core, bounded sample, rank 0 ---
    cpu = None
    if rank == 0 and not a.no_cpu_baseline:
        from oracle import oracle as O
        m = min(n, a.cpu_sample)
        xs = x_host[:m]
        tc = td = 0.0
        for _ in range(max(1, a.cpu_repeats)):
            c0 = time.perf_counter()
            c = O.compress(xs, a.eb, O.QT if a.mode == "qt" else O.EC, O.FAST)
            c1 = time.perf_counter()
            O.decompress(c, O.FAST)
            c2 = time.perf_counter()
            tc += c1 - c0
            td += c2 - c1
        reps = max(1, a.cpu_repeats)
        # the same port on several host cores at once ("all cores via one process per shard"):
        # contiguous block-aligned slices of the sample, one thread each (the C oracle runs outside the GIL)
        from concurrent.futures import ThreadPoolExecutor
        nthr = max(1, min(16, os.cpu_count() or 1))
        cuts = [(m * i // nthr) // 64 * 64 for i in range(nthr)] + [m]

        def one(i):
            sl = xs[cuts[i]:cuts[i + 1]]
            if sl.size:
                O.decompress(O.compress(sl, a.eb, O.QT if a.mode == "qt" else O.EC, O.FAST), O.FAST)

        with ThreadPoolExecutor(nthr) as ex:
            m0 = time.perf_counter()
            list(ex.map(one, range(nthr)))
            tm = time.perf_counter() - m0
        cpu = {"value": reps * m * es / (tc + td) / 1e9, "unit": "GB/s (input bytes, compress+decompress)",
               "cores": 1, "kind": "port",
               "sample": f"first {m} elements of the rank-0 shard ({m * es / 2**20:.0f} MiB) x {reps} passes, oracle FAST "
                         f"flow, compress {tc:.2f} s + decompress {td:.2f} s in all; zlib excluded on both sides",
               "compress_GBps": reps * m * es / tc / 1e9, "decompress_GBps": reps * m * es / td / 1e9,
               "multi_core": {"cores": nthr, "value": m * es / tm / 1e9,
                              "note": "same port, one pass, the sample cut into one slice per thread (each slice its own sf)"}}

    if rank == 0:
        value = n * es * a.gpus / (ms_per_step * 1e-3) / 1e9
        line = {
            "metric": "compress+decompress GB/s (input bytes), fp64 1e-3 EC" if (a.dtype == "f64" and a.mode == "ec")
                      else f"compress+decompress GB/s (input bytes), {a.dtype} {a.eb:g} {a.mode.upper()}",
            "value": value, "unit": "GB/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"C4 shard per GPU: synthetic {a.dtype} {a.n}^3 volume (C3 formula, seed 512+rank), "
                                   f"{a.mode.upper()} eb={a.eb:g}; step = dctzhip_compress + dctzhip_decompress, "
                                   "inputs resident in HBM",
                       "elements_per_gpu": n, "exception_fraction": p, "parallelism": f"shard-per-gpu x{a.gpus}"},
            "pct_hbm_peak_input": 100.0 * (n * es / (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBPS,
            "roofline": {"bound": "hbm", "kernel": "k_compress (fused scale+DCT-II+binning+ordered AC_exact" + ("+max/min/sum)" if fused else ")"),
                         "achieved": ach_c, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach_c / HBM_PEAK_GBPS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": bytes_c_main,
                         "avg_launch_ms": acc["c_main"]},
            "statistics": ("fused into k_compress behind a sampled guess of sf, verified every step "
                           "(k_stats below = the 1/64 sample + final reduction)") if fused else "separate k_stats pass",
            "kernels": {"k_stats": ({"ms": acc["c_stats"], "sampled_fraction": 1.0 / 64.0} if fused else
                                    {"ms": acc["c_stats"], "GBps": ach_s, "frac": ach_s / HBM_PEAK_GBPS}),
                        "k_compress": {"ms": acc["c_main"], "GBps": ach_c, "frac": ach_c / HBM_PEAK_GBPS},
                        "k_decompress": {"ms": acc["d_main"], "GBps": ach_d, "frac": ach_d / HBM_PEAK_GBPS},
                        "compress_tail_ms": acc["c_tail"], "decompress_tail_ms": acc["d_tail"]},
            "host_call_ms": {"compress": t_c, "decompress": t_d},
            "compress_GBps_input": n * es / (t_c * 1e-3) / 1e9,
            "decompress_GBps_input": n * es / (t_d * 1e-3) / 1e9,
            "cpu_baseline": cpu,
        }
        if gather_ms is not None:
            line["gather_ms"] = gather_ms
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
