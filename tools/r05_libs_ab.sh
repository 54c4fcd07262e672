#!/bin/bash
# headline bench, builds alternating on one box:  bash tools/r05_libs_ab.sh TAG name=libdir [name=libdir ...]   (libdir under dctz_amd/)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for rep in $(seq 1 ${REPS:-3}); do for v in "$@"; do
  name=${v%%=*}; dir=${v#*=}
  DCTZHIP_LIBRARY=$R/dctz_amd/$dir/libdctzhip.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage --steps 20 --warmup 5 ${BENCH_ARGS:-} > $O/${name}_$rep.json 2> $O/${name}_$rep.err || echo "run $name $rep failed"
done; done
python3 - $O "$@" <<'PY'
import json, sys, glob, statistics
o = sys.argv[1]
for v in sys.argv[2:]:
    name = v.split("=")[0]
    rows = []
    for f in sorted(glob.glob(f"{o}/{name}_[0-9].json")):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        rows.append((d["kernels"]["k_compress"]["ms"], d["kernels"]["k_decompress"]["ms"], d["settled_ms_per_step"]))
    if rows:
        print(name.ljust(10), "k_compress", [round(r[0], 4) for r in rows], "median", round(statistics.median(r[0] for r in rows), 4), " settled step", round(statistics.median(r[2] for r in rows), 4))
PY
