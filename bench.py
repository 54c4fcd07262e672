#!/usr/bin/env python3
"""bench.py -- DCTZ hot path on MI355X: compress + decompress throughput.

One "step" = one dctzhip_compress() followed by one dctzhip_decompress() of one
device-resident shard (the block-DCT + binning path of dctz_compress /
dctz_decompress; zlib tail excluded -- it stays on the host).

Workload (BASELINE.json metric "fp64 1e-3 EC"): one C4 shard per GPU = synthetic
fp64 512^3 volume (C3 formula, seed 512+rank), EC mode, error bound 1e-3.
Weak scaling: every rank owns its shard, no data-path collective (shards are
independent dctz_compress calls).

Ranks.  `--gpus N` with N > 1 needs N processes, one per GPU:
  * launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
    (RANK / LOCAL_RANK / WORLD_SIZE in the environment): this process IS a rank;
  * launched plainly (`python bench.py --gpus N`): this process touches no GPU, starts N
    children (itself, with the rank environment set), forwards rank 0's JSON line and
    exits non-zero if any child fails.
A rank whose world size differs from --gpus exits non-zero: a number is never scaled by a
GPU count that did not run.  The line carries "ranks_seen" and every rank's device.
For N > 1 a second timed region puts the one real exchange step -- the gather of the
pre-zlib streams to rank 0 over RCCL (dctzhip_comm_gather) -- INSIDE the step
("with_gather").

`--plumbing-only`: the launcher, the rendezvous and the gather alone (gloo, synthetic byte
streams, no kernels, no GPU, no `value`): what the CPU test of the N > 1 path runs.

`--config c1|c2|c3|c4|c5` selects one of BASELINE.json's five configurations at its own size (default c4 = the headline:
the line's shape is the same for all of them).  c5 is a LIST of arrays (the six list-msst19 lengths in fp64 and the five
CESM-sized fp32 fields, error bounds 1e-3 .. 1e-6 each): its step is one dctzhip_compress_batch + one
dctzhip_decompress_batch over the whole list, and the line also carries the same list done one call per array ("looped").

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def kernel_source_hash(root=ROOT):
    """sha256 over the sources every kernel of the library is built from (dctz_amd/csrc/*.hip|*.h, include/*.h, the build
    flags in dctz_amd/Makefile): what ties a committed PMC traffic record to the code it was measured on.  (The GPU box
    has no .git, and a rebuilt .so differs in more than its kernels.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "dctz_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "dctz_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(root, "include", "*.h")) + [os.path.join(root, "dctz_amd", "Makefile")])
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def lookup_traffic(record_path, kernel, key, source_hash):
    """(traffic, source) from a committed PMC record, or (None, reason): a record taken on other sources is never quoted."""
    try:
        rec = json.load(open(record_path))
    except (OSError, ValueError):
        return None, None
    name = os.path.relpath(record_path, ROOT)
    if rec.get("source_hash") != source_hash:
        return None, f"{name} was measured on sources {rec.get('source_hash')}, this build is {source_hash}: not quoted"
    ent = rec.get("kernels", {}).get(kernel, {}).get(key)
    if ent is None:
        return None, None
    return ent["hbm_bytes_per_launch"], name + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on these sources, gfx950 correction applied)"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (0.5 ms each: the default is a tenth of a second of GPU time)")
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps first (clocks and caches settle over the first dozens of steps)")
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c5"], default="c4",
                    help="BASELINE.json configuration (c4 = the headline shard; --n / --eb / --mode / --dtype vary c3 / c4 only)")
    ap.add_argument("--n", type=int, default=512, help="volume edge (512 -> 1 GiB fp64 shard)")
    ap.add_argument("--eb", type=float, default=None)
    ap.add_argument("--mode", choices=["ec", "qt"], default=None)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--settle-ms", type=float, default=25.0,
                    help="untimed steps of the same workload for this long before the warm-up steps (allocations, clocks); 0 = none")
    ap.add_argument("--plumbing-only", action="store_true", help="launcher + rendezvous + gather on gloo; no kernels, no value")
    ap.add_argument("--launch-timeout", type=float, default=480.0, help="the launcher ends all ranks after this many seconds (below the driver's own limit)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 ranks that SHARE cuda:0 (gloo for the barriers, tests/c/librccl_double.so in RCCL's place): the whole "
                         "multi-rank code path on a one-GPU box; the line says so and its number is not a scaling measurement")
    ap.add_argument("--gather-timeout", type=float, default=150.0, help="N > 1: seconds the (last) gather phase may take before the line is printed without it")
    ap.add_argument("--plumbing-fail-rank", type=int, default=-1, help="(tests) this rank of a --plumbing-only run exits with code 7 before the rendezvous")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-entropy-stage", action="store_true", help="skip the (untimed) report on the device entropy stage")
    ap.add_argument("--no-speculation", action="store_true", help="always run the separate statistics pass first")
    ap.add_argument("--cpu-sample", type=int, default=1 << 27, help="elements of the shard the CPU oracle is timed on")
    ap.add_argument("--cpu-repeats", type=int, default=4, help="passes of the CPU oracle over the sample (about 10 s in all)")
    a = ap.parse_args(argv)
    if a.mode is None:
        a.mode = "qt" if a.config == "c3" else "ec"
    if a.eb is None:
        a.eb = 1e-4 if a.config == "c2" else 1e-3
    if a.config == "c2":
        a.dtype = "f32"
    elif a.config in ("c1", "c5"):
        a.dtype = "f64"
    return a


def workload(a, rank):
    """The arrays of one rank for --config: (name, [host arrays], [error bounds])."""
    import numpy as np
    from tests import workloads as W
    if a.config == "c1":
        return ("C1: 2^20 uniform[0,1) doubles (default_rng(12345+rank)), EC eb=1e-3 (91.5 % of the coefficients stored exactly)",
                [np.random.default_rng(12345 + rank).random(1 << 20)], [a.eb])
    if a.config == "c2":
        return ("C2: CESM-ATM stand-in, smooth 1800x3600 fp32 field (seed 2024+rank), EC eb=1e-4", [W.c2(2024 + rank)], [a.eb])
    if a.config == "c5":
        xs, ebs = [], []
        for i, n in enumerate(W.MSST19_LENGTHS):
            for eb in (1e-3, 1e-4, 1e-5, 1e-6):
                xs.append(W.c5_fp64(n, 100 + i + 1000 * rank)); ebs.append(eb)
        for f in range(5):                                     # tests/list-CESM-ATM-tylor.txt:1-5: five 3600 x 1800 fp32 fields
            x = W.c2(3000 + f + 1000 * rank)
            for eb in (1e-3, 1e-4, 1e-5, 1e-6):
                xs.append(x); ebs.append(eb)
        return ("C5: list-msst19 lengths (6 fp64 arrays of 12 960 .. 37 024) + 5 CESM-sized fp32 fields (1800x3600), "
                f"eb 1e-3 .. 1e-6 each = {len(xs)} arrays, {a.mode.upper()}; step = ONE dctzhip_compress_batch + ONE dctzhip_decompress_batch",
                xs, ebs)
    np_dtype = np.float64 if a.dtype == "f64" else np.float32
    tag = "C3" if a.config == "c3" else "C4 shard per GPU"
    return (f"{tag}: synthetic {a.dtype} {a.n}^3 volume (C3 formula, seed 512+rank), {a.mode.upper()} eb={a.eb:g}",
            [W.c3(a.n, seed=512 + rank, dtype=np_dtype)], [a.eb])


# exit codes of a rank whose LAST phase (the gather of an N-rank run) failed or timed out: rank 0 has printed its line
RC_GATHER_PEER, RC_GATHER_RANK0 = 3, 4


def _complete_line(raw):
    """rank 0's stdout if it holds one complete JSON result line, else None"""
    txt = (raw or b"").decode(errors="replace")
    for ln in reversed(txt.strip().splitlines()):
        try:
            d = json.loads(ln)
        except ValueError:
            continue
        if isinstance(d, dict) and ("metric" in d or "plumbing_only" in d):
            return ln + "\n"
    return None


# ------------------------------------------------------------------ launcher --
def launch(a):
    """Parent of an N-rank run.  Makes NO GPU call (imports neither the array framework nor the library): it only
    starts the ranks -- fresh child processes, never a re-exec -- and watches them: the first rank that exits non-zero
    (or the deadline, --launch-timeout seconds) ends the others within seconds, so that a rank that dies at start-up
    cannot leave its peers sitting in the rendezvous until the collective library's own time-out."""
    pre = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCPROFILER_LIBRARY", "ROCP_TOOL_LIBRARIES", "ROCPROF_ATTACH_TOOL_LIBRARY"))
    if "rocprofiler" in pre or "rocprof" in pre:
        # the profiler's preload has initialised the GPU in THIS process: starting children from it is the exec hop the
        # pool forbids.  Profile one rank: rocprofv3 ... -- python3 bench.py --gpus 1 (tools/README.md)
        print("bench.py: refusing to launch ranks from under a rocprof preload; profile a single rank instead", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", DCTZ_BENCH_LAUNCHED="1")
        if a.rehearse_one_gpu:
            env["DCTZHIP_RCCL_LIBRARY"] = os.environ.get("DCTZ_BENCH_RCCL_OVERRIDE") or \
                os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "c", "librccl_double.so")   # (the override: tests of the failure path)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + a.launch_timeout
    why = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0, RC_GATHER_PEER, RC_GATHER_RANK0)]
        left = [i for i, rc in enumerate(rcs) if rc in (RC_GATHER_PEER, RC_GATHER_RANK0)]
        if left and not bad:
            # a rank has left from the gather phase (the last one): rank 0's line is out or about to be -- give the others a
            # few seconds to follow instead of ending them at once
            deadline = min(deadline, time.monotonic() + 10.0)
            if time.monotonic() > deadline - 0.01 and why is None:
                why = f"rank {left[0]} left the gather phase with code {rcs[left[0]]}"
        if bad:
            why = f"rank {bad[0]} exited with code {rcs[bad[0]]}"
        elif time.monotonic() > deadline:
            why = f"no result after {a.launch_timeout:.0f} s"
        if why:
            for p in procs:                                    # the exact children started above, nothing else
                if p.poll() is None:
                    p.terminate()
            t_kill = time.monotonic() + 5.0
            while any(p.poll() is None for p in procs) and time.monotonic() < t_kill:
                time.sleep(0.05)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=5.0)
    if why is None and not any(rcs):
        sys.stdout.write((out0[0] if out0 else b"").decode())
        sys.stdout.flush()
        return 0
    # A run whose LAST phase failed has measured everything its line reports: a complete line of rank 0 is forwarded
    # whatever happened afterwards (the line's with_gather says what), the failing ranks go to stderr, and the exit code is
    # 0 only if rank 0 itself ended in one of the two ways that follow a printed line.
    line = _complete_line(out0[0] if out0 else b"")
    print(f"bench.py: {why or 'a rank failed'}; rank exit codes {rcs}", file=sys.stderr)
    if line is not None and rcs[0] in (0, RC_GATHER_RANK0):
        sys.stdout.write(line)
        sys.stdout.flush()
        return 0
    return 1


# ------------------------------------------------------------ plumbing only --
def plumbing(a, rank, world):
    """Rendezvous + the stream gather on gloo with synthetic byte streams: no kernels, no oracle, no GPU."""
    if rank == a.plumbing_fail_rank:
        sys.exit(7)                                           # (tests: a rank that dies at start-up)
    import numpy as np
    import torch
    import torch.distributed as dist
    from dctz_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if dist.get_world_size() != a.gpus:
        print(f"bench.py: world size {dist.get_world_size()} != --gpus {a.gpus}", file=sys.stderr)
        sys.exit(2)
    rng = np.random.default_rng(1000 + rank)
    n = 64 * (50 + 7 * rank) + 13 * rank                      # ragged, different per rank
    nblk = (n + 63) // 64
    cnt = int(rng.integers(1, n // 2))
    streams = {"bin_index": torch.from_numpy(rng.integers(0, 256, n, dtype=np.uint8)),
               "dc": torch.from_numpy(rng.standard_normal(nblk).astype(np.float32)),
               "ac_exact": torch.from_numpy(rng.standard_normal(cnt + 3).astype(np.float32))}

    def digest(s, c):
        return [int(s["bin_index"].to(torch.int64).sum()), float(s["dc"].double().sum()), float(s["ac_exact"][:c].double().sum())]

    mine = [rank, n, cnt] + digest(streams, cnt)
    seen = [None] * world
    dist.all_gather_object(seen, mine)
    got = shard.gather_streams(streams, cnt, dst=0)
    ok = True
    if rank == 0:
        ok = len(got) == world
        for r in range(world):
            ok = ok and got[r]["n"] == seen[r][1] and got[r]["cnt"] == seen[r][2] and digest(got[r], got[r]["cnt"]) == seen[r][3:]
        print(json.dumps({"plumbing_only": True, "n_gpus": a.gpus, "ranks_seen": len({s[0] for s in seen}), "backend": "gloo",
                          "gather_ok": bool(ok), "shards": [{"rank": s[0], "n": s[1], "cnt": s[2]} for s in seen]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 3


# ----------------------------------------------------------------- one rank --
def run_rank(a, rank, local_rank, world):
    import numpy as np
    import torch
    import dctz_amd
    from dctz_amd import shard

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.rehearse_one_gpu:
            if world > 6:
                print("bench.py: --rehearse-one-gpu takes at most 6 ranks (processes on one card)", file=sys.stderr)
                sys.exit(2)
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    seen_world = dist.get_world_size() if dist is not None else 1
    if seen_world != a.gpus:
        print(f"bench.py: world size {seen_world} != --gpus {a.gpus}: refusing to report a scaled number", file=sys.stderr)
        sys.exit(2)
    dev = local_rank if world > 1 else 0
    ctx = dctz_amd.Context(dev)
    devices = [None] * world
    me = {"rank": rank, "device": f"cuda:{dev}", "name": torch.cuda.get_device_name(dev), "pid": os.getpid()}
    if dist is not None:
        dist.all_gather_object(devices, me)
    else:
        devices = [me]

    mode = dctz_amd.QT if a.mode == "qt" else dctz_amd.EC
    qt = mode == dctz_amd.QT
    wl_name, xs_host, ebs = workload(a, rank)
    many = len(xs_host) > 1                                # a list of arrays: the batch entry points
    tdt = [torch.float64 if x.dtype == np.float64 else torch.float32 for x in xs_host]
    uniq = {}                                              # (the same field under several bounds is uploaded once)
    xs = []
    for x in xs_host:
        if id(x) not in uniq:
            uniq[id(x)] = torch.from_numpy(x).to(ctx.device)
        xs.append(uniq[id(x)])
    ns = [x.size for x in xs_host]
    in_bytes = sum(x.size * x.itemsize for x in xs_host)   # input bytes of one step on this rank
    outs = [ctx.alloc_outputs(n) for n in ns]
    recs = [torch.empty(n, dtype=d, device=ctx.device) for n, d in zip(ns, tdt)]
    if not many:
        ctx.reserve(ns[0], tdt[0], mode)
    if a.no_speculation:
        ctx.set_speculation(False)
    x, out, rec, n, es = xs[0], outs[0], recs[0], ns[0], xs_host[0].itemsize
    state = {}

    # (the two C-ABI calls of a step with their arguments converted once: a dozen tensor-attribute look-ups and a fresh info
    # structure per call cost the Python side several microseconds -- a tenth of a C1 / C2 step -- that a C caller's loop
    # does not have)
    pair = None if many else ctx.prepare_pair(x, out, rec, ebs[0], mode)

    def step():
        if not many:
            info = ctx.compress_prepared(pair)
            ctx.decompress_prepared(pair)
            return [info]
        _, infos, state["cp"] = ctx.compress_batch(xs, ebs, mode, outs=outs, prepared=state.get("cp"))
        if "dp" not in state:                              # (the list is compressed again every step to the same streams)
            _, _, state["dp"] = ctx.decompress_batch(outs, [i.cnt for i in infos], ns, tdt, ebs, [i.sf for i in infos], mode,
                                                     qtables=[np.array(i.qtable[:]) for i in infos] if qt else None, dsts=recs)
        else:
            ctx.decompress_batch(None, None, None, None, None, None, mode, prepared=state["dp"])
        return infos

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    red_dev = "cpu" if a.rehearse_one_gpu else ctx.device   # (gloo reduces host tensors)
    # ---- first the form exactly as asked, nothing in front of it: W warm-up steps of a fresh process, then K timed steps.
    # This is what a caller who makes W + K calls sees (`unsettled_ms_per_step`); the settled figure follows in the same run.
    infos = None
    for _ in range(a.warmup):
        infos = step()
    barrier()
    u0 = time.perf_counter()
    for _ in range(a.steps):
        infos = step()
    barrier()
    unsettled_ms = shard.max_over_ranks(time.perf_counter() - u0, red_dev) * 1e3 / a.steps
    # Settling, in front of a second set of W warm-up steps: the part's power management answers sustained load with a dip
    # -- steps 4 to 12 of a fresh process run 12 % slower than steps 1 to 3 and than everything from step 15 on
    # (tools/warm_probe.py: 0.51 / 0.60 / 0.52 ms; host time follows the GPU's, it is not the library).  The 20-step window
    # behind 5 warm-up steps above measures that dip; the window below measures the steady state.  --settle-ms 0: off.
    settle_steps = 0
    if a.settle_ms > 0:
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        while (time.perf_counter() - s0) * 1e3 < a.settle_ms or settle_steps < 2:
            infos = step()
            torch.cuda.synchronize()
            settle_steps += 1
    for _ in range(a.warmup):
        infos = step()
    if infos is None:
        infos = step()

    # ---- timed region: exactly K steps, profiling off (no event overhead) ------
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        infos = step()
    barrier()
    local_elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(local_elapsed, red_dev)
    settled_ms = elapsed * 1e3 / a.steps
    # The line's `value` / `ms_per_step` is the form EXACTLY AS ASKED (W warm-up steps of a fresh process, K timed steps:
    # ADVICE r4); the figure behind the settling is the secondary one (`settled_ms_per_step` / `settled_value`).
    ms_per_step = unsettled_ms
    info = infos[0]

    # ---- the step with the reference's in-place scaling made visible: dctz_compress divides the CALLER's array by sf
    # (dctz-comp-lib.c:193-216); the device entry point does that on request into d_scaled -- written by the compress kernel
    # itself (the SC variant of k_compress / k_compress_one: the scaled tile goes registers -> LDS image -> rows while it is
    # in registers anyway).  `value` is quoted without it ("add s"); this is the same K steps with it ----
    with_scaled = None
    if not many:
        sc = torch.empty_like(x)
        def step_scaled():
            _, inf = ctx.compress(x, ebs[0], mode, out=out, scaled=sc)
            ctx.decompress(out, inf.cnt, n, tdt[0], ebs[0], inf.sf, mode, qtable=inf.qtable if qt else None, dst=rec)
        for _ in range(3):
            step_scaled()
        barrier()
        w0 = time.perf_counter()
        for _ in range(a.steps):
            step_scaled()
        barrier()
        w_ms = shard.max_over_ranks(time.perf_counter() - w0, red_dev) * 1e3 / a.steps
        with_scaled = {"ms_per_step": w_ms, "value": in_bytes * world / (w_ms * 1e-3) / 1e9,
                       "note": "compress writes x / sf into a second buffer as well (the reference's in-place scaling of the caller's array), by the SC variant of the compress kernel"}
        del sc

    # ---- a list of arrays: the same list with one call per array (what the batch entry points replace) ----
    looped = None
    if many:
        def loop_step():
            for j in range(len(xs)):
                _, ij = ctx.compress(xs[j], ebs[j], mode, out=outs[j])
                ctx.decompress(outs[j], ij.cnt, ns[j], tdt[j], ebs[j], ij.sf, mode, qtable=ij.qtable if qt else None, dst=recs[j])
        loop_step()
        k_loop = max(3, min(a.steps, 20))
        barrier()
        l0 = time.perf_counter()
        for _ in range(k_loop):
            loop_step()
        barrier()
        l_ms = shard.max_over_ranks(time.perf_counter() - l0, red_dev) * 1e3 / k_loop
        looped = {"ms_per_step": l_ms, "value": in_bytes * world / (l_ms * 1e-3) / 1e9, "steps": k_loop, "speedup_of_the_batch": l_ms / ms_per_step,
                  "note": "the same arrays, one dctzhip_compress + one dctzhip_decompress call per array"}

        # the launch-bound part of the list by itself: the 24 small fp64 arrays, and those + ONE fp32 field at C2's bound
        # (the list a single `tests/test-dctz.sh` + one CESM run amounts to), batch against one call per array
        def sub(sel):
            sx, se, so, sr = [xs[j] for j in sel], [ebs[j] for j in sel], [outs[j] for j in sel], [recs[j] for j in sel]
            sn, sd = [ns[j] for j in sel], [tdt[j] for j in sel]
            _, si, cp = ctx.compress_batch(sx, se, mode, outs=so)
            _, _, dp = ctx.decompress_batch(so, [i.cnt for i in si], sn, sd, se, [i.sf for i in si], mode,
                                            qtables=[np.array(i.qtable[:]) for i in si] if qt else None, dsts=sr)
            def b_step():
                ctx.compress_batch(None, None, mode, prepared=cp)
                ctx.decompress_batch(None, None, None, None, None, None, mode, prepared=dp)
            def l_step():
                for j in sel:
                    _, ij = ctx.compress(xs[j], ebs[j], mode, out=outs[j])
                    ctx.decompress(outs[j], ij.cnt, ns[j], tdt[j], ebs[j], ij.sf, mode, qtable=ij.qtable if qt else None, dst=recs[j])
            res = {}
            for nm, fn, reps in (("batch", b_step, max(20, a.steps)), ("looped", l_step, max(5, min(a.steps, 20)))):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                q0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                res[nm + "_ms"] = (time.perf_counter() - q0) * 1e3 / reps
            sb = sum(ns[j] * xs_host[j].itemsize for j in sel)
            res.update({"arrays": len(sel), "bytes": int(sb), "speedup_of_the_batch": res["looped_ms"] / res["batch_ms"],
                        "batch_GBps": sb / (res["batch_ms"] * 1e-3) / 1e9, "looped_GBps": sb / (res["looped_ms"] * 1e-3) / 1e9})
            return res
        small = [j for j in range(len(xs)) if xs_host[j].itemsize == 8]
        c2_like = [j for j in range(len(xs)) if xs_host[j].itemsize == 4 and ebs[j] == 1e-4][:1]
        looped["subsets"] = {"msst19_24_small_fp64": sub(small), "msst19_24_plus_one_fp32_field_at_1e-4": sub(small + c2_like)}

    # ---- per-kernel durations (HIP events on the launch stream), same K steps --
    ctx.set_profiling(True)
    acc = {"c_stats": 0.0, "c_main": 0.0, "c_tail": 0.0, "d_pre": 0.0, "d_main": 0.0, "d_tail": 0.0}
    seqs = {"f64": dict(acc), "f32": dict(acc)}                # a list: per element-type launch sequence
    t_c = t_d = 0.0
    for _ in range(a.steps):
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        if many:
            _, infos, _ = ctx.compress_batch(None, None, mode, prepared=state["cp"])
        else:
            _, info = ctx.compress(x, ebs[0], mode, out=out)
        s1 = time.perf_counter()
        if many:
            bt = ctx.batch_timings()
            for nm in seqs:
                seqs[nm]["c_stats"] += bt[nm]["stats_ms"]; seqs[nm]["c_main"] += bt[nm]["main_ms"]; seqs[nm]["c_tail"] += bt[nm]["tail_ms"]
        else:
            tm = ctx.timings()
            acc["c_stats"] += tm["stats_ms"]; acc["c_main"] += tm["main_ms"]; acc["c_tail"] += tm["tail_ms"]
        s2 = time.perf_counter()
        if many:
            ctx.decompress_batch(None, None, None, None, None, None, mode, prepared=state["dp"])
        else:
            ctx.decompress(out, info.cnt, n, tdt[0], ebs[0], info.sf, mode, qtable=info.qtable if qt else None, dst=rec)
        s3 = time.perf_counter()
        if many:
            bt = ctx.batch_timings()
            for nm in seqs:
                seqs[nm]["d_pre"] += bt[nm]["stats_ms"]; seqs[nm]["d_main"] += bt[nm]["main_ms"]; seqs[nm]["d_tail"] += bt[nm]["tail_ms"]
        else:
            tm = ctx.timings()
            acc["d_pre"] += tm["stats_ms"]; acc["d_main"] += tm["main_ms"]; acc["d_tail"] += tm["tail_ms"]
        t_c += s1 - s0; t_d += s3 - s2
    ctx.set_profiling(False)
    info = infos[0] if many else info

    # algorithmic bytes per element: transform pass of compress reads s, writes 1 (bin) + 4/64 (DC) + 4p
    # (AC_exact); decompress reads 1 + 4/64 + 4p, writes s.  A list: the arrays of the element type that carries most
    # of the list's bytes (its launch sequence holds the dominant kernel)
    def alg_bytes(js):
        return sum(ns[j] * (xs_host[j].itemsize + 1.0 + 4.0 / 64.0 + 4.0 * infos_all[j].cnt / ns[j]) for j in js)
    infos_all = infos if many else [info]
    kern_name = {"c": "k_compress", "d": "k_decompress"}
    one_launch = bool(infos_all[0].flags & dctz_amd.hip.INFO_ONE_LAUNCH)   # the whole call was ONE kernel (dctz_kernels_one.hip)
    if one_launch and not many:
        tn = "double" if es == 8 else "float"
        kern_name = {"c": f"k_compress_one<{tn}>", "d": f"k_decompress_one<{tn}>"}
    if many:
        by = {"f64": [j for j in range(len(xs)) if xs_host[j].itemsize == 8], "f32": [j for j in range(len(xs)) if xs_host[j].itemsize == 4]}
        dom_seq = max(by, key=lambda nm: sum(ns[j] * xs_host[j].itemsize for j in by[nm]))
        for nm in seqs:
            for k in seqs[nm]:
                seqs[nm][k] /= a.steps
        acc = dict(seqs[dom_seq])
        bytes_main = alg_bytes(by[dom_seq])
        bytes_stats = sum(ns[j] * xs_host[j].itemsize for j in by[dom_seq])
        kern_name = {"c": f"k_compress_batch<{'double' if dom_seq == 'f64' else 'float'}>", "d": f"k_decompress_batch<{'double' if dom_seq == 'f64' else 'float'}>"}
        other = "f32" if dom_seq == "f64" else "f64"
        kern_all = sum(seqs[nm][k] for nm in seqs for k in seqs[nm])
    else:
        for k in acc:
            acc[k] /= a.steps
        bytes_main = alg_bytes([0])
        bytes_stats = n * es
        kern_all = sum(acc.values())
    t_c = t_c * 1e3 / a.steps
    t_d = t_d * 1e3 / a.steps

    fused = bool(info.flags & dctz_amd.hip.INFO_STATS_FUSED)     # statistics computed inside k_compress (guess verified)
    p = sum(i.cnt for i in infos_all) / float(sum(ns))          # exception fraction (of the whole list)
    ach_c = bytes_main / (acc["c_main"] * 1e-3) / 1e9
    ach_d = bytes_main / (acc["d_main"] * 1e-3) / 1e9
    ach_s = bytes_stats / (acc["c_stats"] * 1e-3) / 1e9
    dominant = "c" if acc["c_main"] >= acc["d_main"] else "d"

    # HBM traffic of the dominant kernel: NOT measured in this run (PMC counters need rocprofv3 passes of their own);
    # the committed record of the same command on the same build is quoted with its source, or null
    src_hash = kernel_source_hash()
    tkey = f"{a.config}_{a.dtype}_{a.n}_{a.mode}_{a.eb:g}"
    # (the record is keyed by the kernel's full name -- every template argument: variants of one kernel differ in traffic)
    kern_exact = {"c": ctx.last_kernel(0) or kern_name["c"], "d": ctx.last_kernel(1) or kern_name["d"]}
    if many:
        kern_exact = {"c": ctx.last_kernel(2 if dom_seq == "f64" else 3) or kern_name["c"], "d": ctx.last_kernel(4 if dom_seq == "f64" else 5) or kern_name["d"]}
    traffic, traffic_source = lookup_traffic(os.path.join(ROOT, "profiles", "pmc_traffic.json"), kern_exact[dominant], tkey, src_hash)

    # ---- the entropy stage on the device, rank 0, outside the timed region: what
    # it costs to turn the streams of the last compress call into the container's three zlib sections in HBM ----
    entropy = None
    if rank == 0 and not a.no_entropy_stage and not many:
        try:
            cnt = int(info.cnt)
            secs = [out["bin_index"], out["dc"], out["ac_exact"][:cnt]]
            raw = [t.numel() * t.element_size() for t in secs]
            zs = ctx.deflate(secs, literals=[False, True, True])
            torch.cuda.synchronize()
            tt = []
            for _ in range(10):
                e0 = time.perf_counter()
                zs = ctx.deflate(secs, literals=[False, True, True])   # returns after the stream has drained (it hands the lengths back)
                tt.append(time.perf_counter() - e0)
            med = sorted(tt)[len(tt) // 2]
            entropy = {"what": "dctzhip_deflate of bin_index / DC / AC_exact where k_compress left them: three standard zlib streams in HBM",
                       "ms": med * 1e3, "raw_bytes": sum(raw), "stream_bytes": int(sum(z.numel() for z in zs)),
                       "GBps_of_streams": sum(raw) / med / 1e9, "container_ratio": n * es / float(sum(z.numel() for z in zs) + 56),
                       "note": "not part of `value`; reference: three host threads of zlib, 6.6 s per GiB shard (profiles/r02_e2e_dropin.json)"}
        except Exception as ex:                            # the stage is reported, never required, by the benchmark
            entropy = {"error": str(ex)}

    # ---- CPU baseline: the oracle (a port), bounded sample, rank 0 only ---
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import oracle as O
        omode = O.QT if a.mode == "qt" else O.EC
        # the sample: the rank-0 arrays themselves (each bounded by --cpu-sample elements), passed over until about
        # ten seconds of CPU work are on the clock (a 1 GiB shard: --cpu-repeats passes)
        samp = [xh[:min(xh.size, a.cpu_sample)] for xh in xs_host]
        s_bytes = sum(v.size * v.itemsize for v in samp)
        tc = td = 0.0
        reps = 0
        while True:
            for v, eb in zip(samp, ebs):
                c0 = time.perf_counter()
                c = O.compress(v, eb, omode, O.FAST)
                c1 = time.perf_counter()
                O.decompress(c, O.FAST)
                c2 = time.perf_counter()
                tc += c1 - c0
                td += c2 - c1
            reps += 1
            if (reps >= max(1, a.cpu_repeats) and tc + td >= 3.0) or tc + td >= 20.0 or reps >= 4000:
                break
        # the same port on ALL host cores at once ("all cores via one process per shard", core count stated):
        # contiguous block-aligned slices of the sample, one thread each (the C oracle runs outside the GIL); a list of
        # arrays: whole arrays dealt round to the threads
        from concurrent.futures import ThreadPoolExecutor
        nthr = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        if many:
            jobs = [(v, eb) for v, eb in zip(samp, ebs)]
        else:
            m = samp[0].size
            cuts = [(m * i // nthr) // 64 * 64 for i in range(nthr)] + [m]
            jobs = [(samp[0][cuts[i]:cuts[i + 1]], ebs[0]) for i in range(nthr)]

        def one(job):
            if job[0].size:
                O.decompress(O.compress(job[0], job[1], omode, O.FAST), O.FAST)

        with ThreadPoolExecutor(nthr) as ex:
            list(ex.map(one, jobs))                         # warm the threads / page in
            m0 = time.perf_counter()
            list(ex.map(one, jobs))
            tm_all = time.perf_counter() - m0
        cpu = {"value": reps * s_bytes / (tc + td) / 1e9, "unit": "GB/s (input bytes, compress+decompress)",
               "cores": 1, "kind": "port",
               "sample": f"the rank-0 workload ({len(samp)} array(s), {s_bytes / 2**20:.1f} MiB) x {reps} passes, oracle FAST "
                         f"flow, compress {tc:.2f} s + decompress {td:.2f} s in all; zlib excluded on both sides",
               "compress_GBps": reps * s_bytes / tc / 1e9, "decompress_GBps": reps * s_bytes / td / 1e9,
               "all_cores": {"cores": nthr, "host_cpu_count": os.cpu_count(), "value": s_bytes / tm_all / 1e9,
                             "note": "same port, one pass: " + ("whole arrays dealt to the threads" if many else
                                     "the sample cut into one slice per available core (each slice its own sf)")}}

    # every rank's own figures (an N-GPU run: the curve's points come with the spread behind them)
    mine_rank = {"rank": rank, "device": f"cuda:{dev}", "ms_per_step_local": local_elapsed * 1e3 / a.steps,
                 "k_compress_ms": acc["c_main"], "k_decompress_ms": acc["d_main"], "compress_tail_ms": acc["c_tail"],
                 "decompress_count_ms": acc["d_pre"], "exception_fraction": p}
    per_rank = [None] * world
    if dist is not None:
        dist.all_gather_object(per_rank, mine_rank)
    else:
        per_rank = [mine_rank]
    if rank == 0:
        value = in_bytes * world / (ms_per_step * 1e-3) / 1e9
        kern_sum = kern_all
        dom = {"c": (acc["c_main"], ach_c), "d": (acc["d_main"], ach_d)}[dominant]
        headline = a.config == "c4" and a.dtype == "f64" and a.mode == "ec" and a.eb == 1e-3
        line = {
            "metric": "compress+decompress GB/s (input bytes), fp64 1e-3 EC" if headline
                      else f"compress+decompress GB/s (input bytes), config {a.config}: {a.dtype if not many else 'fp64+fp32 list'} "
                           f"{a.mode.upper()}" + ("" if many else f" eb={a.eb:g}"),
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # value / ms_per_step: the K steps behind nothing but the W warm-up steps of a fresh process, what a caller who makes
            # W + K calls sees.  settled_*: the same K steps timed a second time in this run, behind `settle` (untimed steps) and
            # another W warm-up steps -- the steady state; both are of this build and this box.
            "settled_ms_per_step": settled_ms, "settled_value": in_bytes * world / (settled_ms * 1e-3) / 1e9,
            "settle": {"ms": a.settle_ms, "steps": settle_steps, "note": "untimed steps between the as-asked window and the settled one: the "
                       "power management's dip under fresh load (steps 4-12 of a process run 12 % slow, tools/warm_probe.py)"},
            "unsettled_ms_per_step": unsettled_ms, "unsettled_value": in_bytes * world / (unsettled_ms * 1e-3) / 1e9,
            "kernel_source_hash": src_hash,
            "dtype": a.dtype if not many else "f64+f32", "data": "synthetic",
            "config": {"workload": wl_name + ("; step = dctzhip_compress + dctzhip_decompress" if not many else "") + ", inputs resident in HBM",
                       "config": a.config, "arrays_per_gpu": len(xs), "elements_per_gpu": int(sum(ns)), "bytes_per_gpu": int(in_bytes),
                       "exception_fraction": p, "parallelism": f"shard-per-gpu x{world}"},
            "ranks_seen": len({d["rank"] for d in devices}), "devices": devices, "per_rank": per_rank,
            "pct_hbm_peak_input": 100.0 * (in_bytes / (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBPS,
            # kernel: the name rocprofv3 lists the timed kernel under (what the context says it launched last); which: how it was chosen
            "roofline": {"bound": "hbm", "kernel": kern_exact[dominant],
                         "which": "the longer of the two big kernels in THIS run" + (f", launch sequence of the {dom_seq} arrays: they carry most of the list's bytes" if many else ""),
                         "achieved": dom[1], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": dom[1] / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": bytes_main, "avg_launch_ms": dom[0],
                         "k_compress_frac": ach_c / HBM_PEAK_GBPS, "k_decompress_frac": ach_d / HBM_PEAK_GBPS},
            "statistics": ("inside the one kernel of the call: every workgroup posts the decade of its maximum, the scaling factor follows "
                           "from a sweep of those (dctz_kernels_one.hip)") if one_launch else
                          ("fused into k_compress behind a sampled guess of sf, verified every step "
                           "(k_stats below = the 1/64 sample + final reduction)") if fused else "separate k_stats pass",
            "one_launch_per_call": one_launch,
            "kernels": {"k_stats": ({"ms": acc["c_stats"], "sampled_fraction": 1.0 / 64.0} if fused else
                                    {"ms": acc["c_stats"], "GBps": ach_s, "frac": ach_s / HBM_PEAK_GBPS}),
                        "k_compress": {"ms": acc["c_main"], "GBps": ach_c, "frac": ach_c / HBM_PEAK_GBPS},
                        "k_decompress": {"ms": acc["d_main"], "GBps": ach_d, "frac": ach_d / HBM_PEAK_GBPS},
                        "compress_tail_ms": acc["c_tail"], "decompress_count_scan_ms": acc["d_pre"], "decompress_tail_ms": acc["d_tail"],
                        "sum_ms": kern_sum},
            # step time minus the event spans.  The spans come from a SEPARATE, profiled run of the same K steps and carry the
            # event markers' own cost (about 4 us per span); since the host returns from a call while its last kernel is still
            # running (early hand-off) the unprofiled step can be SHORTER than their sum -- the difference is then negative
            "host_gap_ms": ms_per_step - kern_sum,
            "host_call_ms": {"compress": t_c, "decompress": t_d},
            "compress_GBps_input": in_bytes / (t_c * 1e-3) / 1e9,
            "decompress_GBps_input": in_bytes / (t_d * 1e-3) / 1e9,
            "cpu_baseline": cpu,
            "entropy_stage": entropy,
        }
        if many:
            line["kernels"]["other_sequence"] = {"element_type": other, **seqs[other]}
            line["looped"] = looped
        if with_scaled is not None:
            line["with_scaled_copy"] = with_scaled
        if a.rehearse_one_gpu:
            line["rehearsal"] = (f"{world} ranks SHARE one GPU (gloo barriers; the gather runs through the RCCL test double of "
                                 "tests/c/rccl_double.cpp): the N > 1 code path end to end, NOT a scaling measurement")
            line["scaling"] = "none (rehearsal)"
    else:
        line = None

    # ---- N > 1, LAST: the same K steps with the gather of the streams to rank 0 inside the step.  Everything the line
    # reports has been measured by now.  This is the one phase that talks to a second collective library instance (RCCL
    # through the C ABI's own dlopen) and that no box of this project could ever run on N GPUs: if it fails on any rank, or
    # does not finish in --gather-timeout seconds, the line goes out without it (with the reason) instead of not at all.
    if dist is not None and not many:
        import threading
        barrier()                                             # (rank 0 comes from its CPU baseline: the timers below start together)

        out_lock = threading.Lock()                           # the line goes out ONCE: by the watchdog or by the main thread
        emitted = [False]

        def emit(with_gather):
            with out_lock:
                if rank == 0 and not emitted[0]:
                    line["with_gather"] = with_gather
                    print(json.dumps(line), flush=True)
                    emitted[0] = True

        def bail():
            # a hung gather is NOT a clean run: the line (everything it reports was measured before this phase) goes out with
            # the reason, and the exit codes say what happened (the launcher forwards a complete line of rank 0 whatever
            # the codes are)
            emit({"error": f"the gather phase did not finish within {a.gather_timeout:.0f} s"})
            os._exit(RC_GATHER_RANK0 if rank == 0 else RC_GATHER_PEER)
        watchdog = threading.Timer(a.gather_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        with_gather = None
        try:
            ids = [(None, "")]
            if rank == 0:
                try:
                    ids = [(dctz_amd.Context.comm_unique_id(), "")]
                except Exception as e:                          # e.g. RCCL not found by the library's own loader
                    ids = [(None, str(e))]
            dist.broadcast_object_list(ids, src=0)
            mine_ok, why = 1, ""
            if ids[0][0] is None:
                mine_ok, why = 0, ids[0][1] if rank == 0 else "no id from rank 0"
            else:
                try:
                    ctx.comm_create(rank, world, ids[0][0])
                except Exception as e:
                    mine_ok, why = 0, str(e)
            flags = [None] * world
            dist.all_gather_object(flags, (mine_ok, why))
            if all(f[0] for f in flags):
                ctx.comm_gather(out, info.cnt, n, root=0)          # warm-up (communicator set-up, receive buffers)
                barrier()
                g0 = time.perf_counter()
                for _ in range(a.steps):
                    info = step()[0]
                    ctx.comm_gather(out, info.cnt, n, root=0)
                barrier()
                g_ms = shard.max_over_ranks(time.perf_counter() - g0, red_dev) * 1e3 / a.steps
                # ... and the gather by itself (the streams of the last step, K times): what the exchange costs and what rank 0
                # takes in per second over its inbound links
                cnts = [None] * world
                dist.all_gather_object(cnts, int(info.cnt))
                barrier()
                h0 = time.perf_counter()
                for _ in range(a.steps):
                    ctx.comm_gather(out, info.cnt, n, root=0)
                barrier()
                h_ms = shard.max_over_ranks(time.perf_counter() - h0, red_dev) * 1e3 / a.steps
                inbound = sum(n + 4 * ((n + 63) // 64) + 4 * c for r_, c in enumerate(cnts) if r_ != 0)
                with_gather = {"ms_per_step": g_ms, "value": n * es * world / (g_ms * 1e-3) / 1e9,
                               "gather_only_ms": h_ms, "inbound_bytes_per_step": int(inbound),
                               "root_ingest_GBps": inbound / (h_ms * 1e-3) / 1e9 if h_ms > 0 else None,
                               "note": "compress + decompress + RCCL gather of bin_index / DC / AC_exact of every shard to rank 0 per step; "
                                       "gather_only_ms: the gather alone, same streams"}
            else:
                if mine_ok:
                    ctx.lib.dctzhip_comm_destroy(ctx.h)
                with_gather = {"error": "no communicator: " + "; ".join(f"rank {i}: {f[1]}" for i, f in enumerate(flags) if not f[0])}
        except Exception as e:
            with_gather = {"error": f"rank {rank}: {e}"}
        watchdog.cancel()
        emit(with_gather)
        if isinstance(with_gather, dict) and "error" in with_gather:
            # after an error in this phase the ranks are no longer in step: no further collective (a final barrier would
            # hang or fail on the ranks whose peers have left), no clean tear-down of the process group
            if rank != 0:
                print(f"bench.py: rank {rank}: gather phase failed: {with_gather['error']}", file=sys.stderr, flush=True)
            os._exit(RC_GATHER_RANK0 if rank == 0 else RC_GATHER_PEER)
    elif rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return 0


def main():
    a = parse()
    if a.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    env_world = os.environ.get("WORLD_SIZE")
    if a.gpus > 1 and env_world is None:
        return launch(a)                                     # plain invocation: become the launcher (no GPU call here)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != a.gpus:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {a.gpus}: refusing to report a scaled number", file=sys.stderr)
        return 2
    if a.plumbing_only:
        if world == 1:
            print(json.dumps({"plumbing_only": True, "n_gpus": 1, "ranks_seen": 1, "gather_ok": True, "shards": []}), flush=True)
            return 0
        return plumbing(a, rank, world)
    return run_rank(a, rank, local_rank, world)


if __name__ == "__main__":
    sys.exit(main())
