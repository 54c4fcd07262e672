"""bench.py quotes a committed PMC traffic record only for a build of the sources it was measured on (VERDICT r3 #5b:
round 3's line read a hard-coded profiles/r03_pmc_traffic.json whatever the kernels were)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _record(tmp_path, source_hash):
    p = tmp_path / "pmc_traffic.json"
    p.write_text(json.dumps({"source_hash": source_hash,
                             "kernels": {"k_compress": {"c4_f64_512_ec_0.001": {"hbm_bytes_per_launch": 1255610368}}}}))
    return str(p)


def test_matching_record_is_quoted(tmp_path):
    h = bench.kernel_source_hash()
    t, src = bench.lookup_traffic(_record(tmp_path, h), "k_compress", "c4_f64_512_ec_0.001", h)
    assert t == 1255610368 and "pmc_traffic.json" in src


def test_stale_record_is_not_quoted(tmp_path):
    h = bench.kernel_source_hash()
    t, src = bench.lookup_traffic(_record(tmp_path, "0123456789abcdef"), "k_compress", "c4_f64_512_ec_0.001", h)
    assert t is None and "not quoted" in src


def test_round3_record_without_a_hash_is_stale():
    t, src = bench.lookup_traffic(os.path.join(ROOT, "profiles", "r03_pmc_traffic.json"), "k_compress", "c4_f64_512_ec_0.001",
                                  bench.kernel_source_hash())
    assert t is None


def test_missing_record_and_missing_key(tmp_path):
    h = bench.kernel_source_hash()
    assert bench.lookup_traffic(str(tmp_path / "nope.json"), "k_compress", "k", h) == (None, None)
    assert bench.lookup_traffic(_record(tmp_path, h), "k_compress", "another_workload", h) == (None, None)


def test_hash_follows_the_kernel_sources(tmp_path):
    import shutil
    root = tmp_path / "tree"
    for d in ("dctz_amd/csrc", "include"):
        os.makedirs(root / d)
    shutil.copy(os.path.join(ROOT, "dctz_amd", "Makefile"), root / "dctz_amd" / "Makefile")
    (root / "dctz_amd" / "csrc" / "k.hip").write_text("__global__ void k() {}\n")
    (root / "include" / "a.h").write_text("#define A 1\n")
    h0 = bench.kernel_source_hash(str(root))
    assert h0 == bench.kernel_source_hash(str(root))
    (root / "dctz_amd" / "csrc" / "k.hip").write_text("__global__ void k() { }\n")
    assert bench.kernel_source_hash(str(root)) != h0
