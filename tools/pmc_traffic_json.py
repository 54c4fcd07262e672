#!/usr/bin/env python3
"""pmc_traffic.txt (tools/pmc_summary.py over the FETCH_SIZE and the WRITE_SIZE pass) -> the JSON record bench.py
quotes as roofline.traffic:  {"source_hash": ..., "kernels": {kernel: {workload key: {hbm_bytes_per_launch, fetch_kib,
write_kib}}}}.  source_hash = bench.kernel_source_hash() of the tree the passes ran on: bench.py quotes the record only
for a build of the same sources (profiles/pmc_traffic.json is the current record; per-round copies keep their tag).

  python3 tools/pmc_traffic_json.py pmc_traffic.txt KEY [existing.json]     (an existing record of the same sources is extended)

gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KiB, and FETCH_SIZE counts
half of what a 16-byte-per-lane stream reads, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- cross-checked on
k_count_tiles, which reads exactly N bytes (134 217 728) and reports FETCH_SIZE = 65 552 KiB."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402

txt, key = sys.argv[1], sys.argv[2]
vals = {}
for line in open(txt):
    m = re.search(r"dctz::(k_\w+)(<[^>]*>)?.*dispatches \d+ \{'(FETCH_SIZE|WRITE_SIZE)': (\d+)\}", line)
    if m:
        name = m.group(1)
        if name in ("k_compress_one", "k_decompress_one", "k_compress_one_batch", "k_decompress_one_batch", "k_compress_batch", "k_decompress_batch"):
            name = name + "<" + m.group(2).strip("<>").split(",")[0].strip() + ">"     # (bench.py names these with their element type)
        # k_compress<T, MODE, STATS, PH, GEOM, SC>: the SC = true variant also writes the scaled copy (a different kernel for
        # this purpose: 8 bytes per element more)
        if name == "k_compress" and m.group(2) and m.group(2).rstrip(">").split(",")[-1].strip() == "true" and m.group(2).count(",") == 5:
            name = "k_compress_scaled"
        if name == "k_decompress_il":                # (the tile-interleaved form IS the build's k_decompress for fp64 EC: bench.py's name for both)
            name = "k_decompress"
        vals.setdefault(name, {})[m.group(3)] = int(m.group(4))
out = {}
for k, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out[k] = {key: {"hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                        "fetch_size_kib": v["FETCH_SIZE"], "write_size_kib": v["WRITE_SIZE"]}}
h = kernel_source_hash()
rec = {"source_hash": h, "kernels": {}}
if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
    try:
        old = json.load(open(sys.argv[3]))
        if old.get("source_hash") == h:
            rec = old
    except ValueError:
        pass
for k, v in out.items():
    rec["kernels"].setdefault(k, {}).update(v)
json.dump(rec, sys.stdout, indent=1, sort_keys=True)
print()
