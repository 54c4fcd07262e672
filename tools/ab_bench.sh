#!/bin/bash
# A/B of two builds of libdctzhip.so on ONE box, alternating:  bash tools/ab_bench.sh TAG LIB_B [bench args ...]
# (A = the tree's dctz_amd/lib/libdctzhip.so, B = the library given); kernel times are the bench line's HIP-event means
set -u
TAG=$1; LIBB=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rep in 1 2 3; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage "$@" > $O/a$rep.json 2> $O/a$rep.err || exit 1
  DCTZHIP_LIBRARY=$R/$LIBB timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage "$@" > $O/b$rep.json 2> $O/b$rep.err || exit 1
done
python3 - $O <<'PY'
import json, sys, glob
o = sys.argv[1]
for f in sorted(glob.glob(o + "/[ab][0-9].json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = {a: b["ms"] for a, b in d.get("kernels", {}).items() if isinstance(b, dict) and "ms" in b}
    print(f[-7:], "ms/step %.4f" % d["ms_per_step"], {a: round(b, 4) for a, b in k.items()} if isinstance(k, dict) else k)
PY
