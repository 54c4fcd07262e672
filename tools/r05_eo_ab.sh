#!/bin/bash
# A/B on ONE box, alternating: k_compress (default) against k_compress_eo (DCTZHIP_EO=1), headline bench; then a kernel trace
# of the EO run.   bash tools/r05_eo_ab.sh TAG [bench args ...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rep in 1 2 3; do
  DCTZHIP_EO=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage "$@" > $O/a$rep.json 2> $O/a$rep.err || exit 1
  DCTZHIP_EO=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage "$@" > $O/b$rep.json 2> $O/b$rep.err || exit 1
done
python3 - $O <<'PY'
import json, sys, glob
o = sys.argv[1]
for f in sorted(glob.glob(o + "/[ab][0-9].json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = {a: b["ms"] for a, b in d.get("kernels", {}).items() if isinstance(b, dict) and "ms" in b}
    print(f[-7:], "ms/step %.4f unsettled %.4f" % (d["ms_per_step"], d.get("unsettled_ms_per_step", 0)), {a: round(b, 4) for a, b in k.items()})
PY
export DCTZHIP_EO=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --no-entropy-stage "$@" > $O/kt_bench.json 2> $O/kt.err
python3 tools/pmc_summary.py $O/kt > $O/kernel_stats_eo.csv 2>&1
head -12 $O/kernel_stats_eo.csv
