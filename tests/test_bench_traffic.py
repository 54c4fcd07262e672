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


def test_traffic_record_keeps_the_variants_of_a_kernel_apart():
    """round 4's record cut a kernel's name after its first template argument, so the scaled-copy variant of
    k_compress_one (33.8 MB written) overwrote the plain kernel's entry (12.0 MB) -- last line wins -- and the C2 line quoted
    61 MB of traffic for a kernel that moves 39.  The record is keyed by the full name; reads that are 64-byte requests
    (k_decompress_one's bin ids) are not doubled."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic_json as P
    report = """== /x/pf_c2/pf_results.db
void dctz::k_compress_one<float, 0, false> dispatches 139 {'FETCH_SIZE': 13053}
void dctz::k_decompress_one<float, 0> dispatches 162 {'FETCH_SIZE': 8933}
void dctz::k_compress_one<float, 0, true> dispatches 23 {'FETCH_SIZE': 13033}
d dctz::k_compress<double, 0, true, 2, 0, false> dispatches 41 {'FETCH_SIZE': 524000}
== /x/pw_c2/pw_results.db
void dctz::k_compress_one<float, 0, false> dispatches 154 {'WRITE_SIZE': 12006}
void dctz::k_decompress_one<float, 0> dispatches 177 {'WRITE_SIZE': 40189}
void dctz::k_compress_one<float, 0, true> dispatches 23 {'WRITE_SIZE': 33768}
d dctz::k_compress<double, 0, true, 2, 0, false> dispatches 41 {'WRITE_SIZE': 150000}
""".splitlines()
    rec = P.records(P.parse(report), "c2")
    plain, scaled = rec["k_compress_one<float, 0, false>"]["c2"], rec["k_compress_one<float, 0, true>"]["c2"]
    assert plain["hbm_bytes_per_launch"] == (2 * 13053 + 12006) * 1024
    assert scaled["hbm_bytes_per_launch"] == (2 * 13033 + 33768) * 1024 and scaled["hbm_bytes_per_launch"] > plain["hbm_bytes_per_launch"]
    dec = rec["k_decompress_one<float, 0>"]["c2"]
    assert dec["fetch_scale"] == 1 and dec["hbm_bytes_per_launch"] == (8933 + 40189) * 1024
    assert rec["k_compress<double, 0, true, 2, 0, false>"]["c2"]["hbm_bytes_per_launch"] == (2 * 524000 + 150000) * 1024
    assert P.kernel_key("void dctz::k_decompress_il<double,0,1>(dctz::InvParams<double>, dctz::FinArgs) dispatches 3") == "k_decompress_il<double, 0, 1>"
