#!/usr/bin/env python3
"""pmc_traffic.txt (tools/pmc_summary.py over the FETCH_SIZE and the WRITE_SIZE pass) -> the JSON record bench.py
quotes as roofline.traffic:  {"source_hash": ..., "kernels": {kernel: {workload key: {hbm_bytes_per_launch, fetch_kib,
write_kib}}}}.  source_hash = bench.kernel_source_hash() of the tree the passes ran on: bench.py quotes the record only
for a build of the same sources (profiles/pmc_traffic.json is the current record; per-round copies keep their tag).

  python3 tools/pmc_traffic_json.py pmc_traffic.txt KEY [existing.json]     (an existing record of the same sources is extended)

A kernel is keyed by the name rocprofv3 lists it under, EVERY template argument kept ("k_compress_one<float, 0, false>"):
variants of one kernel differ in traffic (the scaled-copy variant of a compress kernel writes 8 bytes per element more),
and round 4's record, which cut the name after its first argument, quoted the variant's bytes for the plain kernel
(VERDICT r4 #8).  bench.py asks for the same name (dctzhip_debug_last_kernel).

gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KiB, and FETCH_SIZE tallies a
128-byte request as 64 bytes, so for the wide streams of the big kernels (16 bytes per lane, whole lines) bytes =
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- cross-checked on k_count_tiles, which reads exactly N bytes (134 217 728) and
reports FETCH_SIZE = 65 552 KiB.  Kernels whose reads are 64-byte requests are NOT doubled: k_decompress_one reads its
bin ids 64 bytes per lane, 16 at a time (TCC_EA0_RDREQ x 64 B = 9.2 MB for 8.6 MB of streams on C2, none of them 32-byte
requests: profiles/r05_c2_raw_traffic.txt); NARROW_READS lists them."""
import json
import os
import re
import sys

NARROW_READS = ("k_decompress_one", "k_decompress_one_batch")


def kernel_key(line):
    """'void dctz::k_compress_one<float, 0, true> dispatches ...' -> 'k_compress_one<float, 0, true>'."""
    m = re.search(r"dctz::(k_\w+)(<[^>]*>)?", line)
    if not m:
        return None
    return m.group(1) + (re.sub(r"\s*,\s*", ", ", m.group(2)) if m.group(2) else "")


def parse(lines):
    """{kernel: {"FETCH_SIZE": KiB, "WRITE_SIZE": KiB}} from the lines of a pmc_summary.py report."""
    vals = {}
    for line in lines:
        m = re.search(r"dispatches \d+ \{'(FETCH_SIZE|WRITE_SIZE)': (\d+)\}", line)
        k = kernel_key(line)
        if m and k:
            vals.setdefault(k, {})[m.group(1)] = int(m.group(2))
    return vals


def records(vals, key):
    out = {}
    for k, v in vals.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fs = 1 if k.split("<")[0] in NARROW_READS else 2
            out[k] = {key: {"hbm_bytes_per_launch": (fs * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024, "fetch_scale": fs,
                            "fetch_size_kib": v["FETCH_SIZE"], "write_size_kib": v["WRITE_SIZE"]}}
    return out


def main(argv):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    txt, key = argv[1], argv[2]
    out = records(parse(open(txt)), key)
    h = kernel_source_hash()
    rec = {"source_hash": h, "kernels": {}}
    if len(argv) > 3 and os.path.exists(argv[3]):
        try:
            old = json.load(open(argv[3]))
            if old.get("source_hash") == h:
                rec = old
        except ValueError:
            pass
    for k, v in out.items():
        rec["kernels"].setdefault(k, {}).update(v)
    json.dump(rec, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main(sys.argv)
