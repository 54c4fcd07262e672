"""dctzhip_comm_sizes + dctzhip_comm_gather (the one exchange step of the multi-GPU path) with REAL peers.

No box of this project has more than one GPU and RCCL refuses two ranks on one device, so the gather had only ever run
with world = 1.  Here `world` processes share cuda:0 and libdctzhip.so loads a test double in RCCL's place
(DCTZHIP_RCCL_LIBRARY = tests/c/librccl_double.so: the same nine entry points, group semantics included, over POSIX shared
memory, device buffers on both ends).  What is checked is the library's own logic: the size exchange, the offsets of every
rank's three streams on the root, any root, repeated calls, and that a failing call inside the group closes the group and
comes back as an error on every rank instead of a hang.  It says nothing about xGMI or RCCL itself."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOUBLE = os.path.join(ROOT, "tests", "c", "librccl_double.so")


def _run(world, root, d, extra_env=None, timeout=240):
    if not os.path.exists(DOUBLE):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "dctz_amd"), "test-doubles"], check=True)
    env = dict(os.environ, DCTZHIP_RCCL_LIBRARY=DOUBLE, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gather_worker.py"), str(r), str(world), str(root), str(d)],
                              env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    t0 = time.monotonic()
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=max(1.0, timeout - (time.monotonic() - t0)))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    return outs, [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world) if os.path.exists(os.path.join(d, f"rank{r}.npz"))]


@pytest.mark.parametrize("world,root", [(2, 0), (3, 1), (4, 3)])
def test_gather_with_peers(world, root, tmp_path):
    outs, recs = _run(world, root, tmp_path)
    assert all(rc == 0 for rc, _, _ in outs), [e[-800:] for _, _, e in outs]
    assert len(recs) == world
    got = recs[root]
    sizes = [(int(r["n"]), (int(r["n"]) + 63) // 64, int(r["cnt"])) for r in recs]
    assert [tuple(int(v) for v in row) for row in got["sizes"]] == sizes          # dctzhip_comm_sizes: every rank's (n, nblk, cnt)
    ob = od = oa = 0
    for r, (n, nb, cn) in zip(recs, sizes):                                       # rank order, back to back
        assert np.array_equal(got["all_bin"][ob:ob + n], r["bin"])
        assert np.array_equal(got["all_dc"][od:od + nb].view(np.uint32), r["dc"].view(np.uint32))
        assert np.array_equal(got["all_ac"][oa:oa + cn].view(np.uint32), r["ac"].view(np.uint32))
        ob += n; od += nb; oa += cn
    assert ob == got["all_bin"].size and od == got["all_dc"].size
    for i, r in enumerate(recs):
        assert ("all_bin" in r.files) == (i == root)                              # nobody else receives anything


def test_a_failing_send_closes_the_group_on_every_rank(tmp_path):
    """a failure between ncclGroupStart and ncclGroupEnd used to return with the group open.  Rank 1's second
    send fails (injected): rank 1 reports it -- after closing its group, which delivers the send that had been queued --
    and the root, which waits for two more messages that never come, gets the double's time-out as an error.  Both come
    back within seconds and exit."""
    t0 = time.monotonic()
    outs, recs = _run(2, 0, tmp_path, {"RCCL_DOUBLE_FAIL": "1:1", "RCCL_DOUBLE_TIMEOUT_S": "3"})
    assert time.monotonic() - t0 < 120
    assert len(recs) == 2 and all(int(r["rc"]) == 1 for r in recs), [(rc, e[-300:]) for rc, _, e in outs]
    assert "ncclSend" in str(recs[1]["err"]) or "Send" in str(recs[1]["err"])
    assert "GroupEnd" in str(recs[0]["err"])


def test_bench_multi_rank_path_rehearsed_on_one_gpu():
    """bench.py's N > 1 path end to end -- launcher, rendezvous, barriers, max over ranks, K steps, the gather step, rank 0's
    one JSON line -- with two ranks that share cuda:0 (--rehearse-one-gpu: gloo + the RCCL double).  The line must say that it
    is a rehearsal: its number is not a scaling measurement."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--n", "256", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--no-entropy-stage"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and [d["rank"] for d in line["devices"]] == [0, 1]
    assert "rehearsal" in line and line["scaling"].startswith("none")
    assert line["with_gather"]["ms_per_step"] > line["ms_per_step"] > 0


def test_bench_line_survives_a_gather_that_cannot_start():
    """The gather is the last phase of an N-rank bench run, and the one nobody could run on N GPUs: when the library finds no
    RCCL to load (here: a name that does not exist) the line must still go out, with the reason in `with_gather`."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DCTZ_BENCH_RCCL_OVERRIDE"] = "/nonexistent/librccl.so"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--n", "256", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--no-entropy-stage"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "error" in line["with_gather"] and "no communicator" in line["with_gather"]["error"]
