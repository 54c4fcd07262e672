#!/bin/bash
# A/B of whole-step time between BUILDS of libdctzhip.so on one box: interleaved runs of bench.py, one process per run.
#   bash tools/ab_libs.sh <out-dir> <rounds> name=path [name=path ...]      (path relative to the repo root)
# Prints ms_per_step and the kernel breakdown of every run, then the per-variant medians.
OUT=$1; ROUNDS=$2; shift 2
mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    name=${v%%=*}; path=${v#*=}
    DCTZHIP_LIBRARY=$PWD/$path python3 bench.py --no-cpu-baseline --steps 40 ${BENCH_ARGS:-} > $OUT/${name}_$r.json 2> $OUT/${name}_$r.err || echo "run $name $r failed"
  done
done
python3 - "$OUT" <<'PY'
import glob, json, os, statistics, sys
out = sys.argv[1]
by = {}
for f in sorted(glob.glob(os.path.join(out, "*_*.json"))):
    name = os.path.basename(f).rsplit("_", 1)[0]
    try:
        d = json.load(open(f))
    except ValueError:
        continue
    k = d["kernels"]
    by.setdefault(name, []).append((d["ms_per_step"], k["k_compress"]["ms"], k["k_decompress"]["ms"], k["sum_ms"]))
for name, rows in by.items():
    med = [round(statistics.median(c), 4) for c in zip(*rows)]
    print(name.ljust(12), "step/compress/decompress/sum ms (median of %d):" % len(rows), med, " steps:", [round(r[0], 4) for r in rows])
PY
