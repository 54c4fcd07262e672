"""bench.py's N > 1 path on the CPU: `python bench.py --gpus 2 --plumbing-only` must start two ranks itself (the parent
makes no GPU call), rendezvous them, gather both shards' streams to rank 0 and say so in its JSON line; a world size
that differs from --gpus must be refused instead of scaling a number by a GPU count that did not run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_launcher_spawns_two_ranks_and_gathers():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["plumbing_only"] is True and "value" not in line
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["gather_ok"] is True
    assert [s["rank"] for s in line["shards"]] == [0, 1] and line["shards"][0]["n"] != line["shards"][1]["n"]


def test_world_size_mismatch_is_refused():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--plumbing-only"],
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""


def test_parent_of_a_multi_rank_run_imports_no_gpu_stack():
    """The launcher path must not initialise HIP (the box forbids exec/fork games from a process that has)."""
    src = open(BENCH).read()
    body = src[src.index("def launch(a):"):src.index("# ------------------------------------------------------------ plumbing only --")]
    assert "import torch" not in body and "dctz_amd" not in body and "cuda" not in body


def test_a_rank_that_dies_at_start_up_ends_the_run_within_seconds():
    """rank 1 exits before the rendezvous; rank 0 would sit in init_process_group until the library's own
    time-out (minutes).  The launcher must end it and exit non-zero, without a JSON line, in seconds."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only", "--plumbing-fail-rank", "1"], env=_env(),
                       capture_output=True, text=True, timeout=120)
    took = time.monotonic() - t0
    assert r.returncode == 1 and "rank 1 exited with code 7" in r.stderr and r.stdout.strip() == "", (r.returncode, r.stderr[-500:])
    assert took < 60, took


def test_launcher_deadline():
    """A run that does not finish by --launch-timeout is ended by the launcher itself (the driver's limit is never reached)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only", "--launch-timeout", "0.01"], env=_env(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no result after" in r.stderr and r.stdout.strip() == ""


def test_launcher_refuses_a_profiler_preload():
    """under rocprofv3 the parent has already initialised the GPU; spawning ranks from it is forbidden."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only"], env=_env(ROCP_TOOL_LIBRARIES="/opt/rocm/lib/librocprofiler-sdk-tool.so"),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "rocprof" in r.stderr and r.stdout.strip() == ""
