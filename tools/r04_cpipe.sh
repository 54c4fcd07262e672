#!/bin/bash
# pipelined dctz_compress: its tests, then the 1 GiB end-to-end figures with the group timeline
set -u
TAG=${1:-r04cpipe}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_libdctz_gpu.py tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or parts_with" > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log
[ $rc -eq 0 ] || exit 1
DCTZ_PIPE_DEBUG=1 timeout -k 10 600 python3 tools/e2e_bench.py --skip-reference-tail --threads 16 > $O/e2e_dropin.json 2> $O/e2e_dropin.err
grep "cpipe" $O/e2e_dropin.err | tail -24
python3 - $O/e2e_dropin.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k, v in d.items():
    if isinstance(v, dict) and "compress_s" in v:
        print(k, "compress %.2f ms  decompress %.2f ms" % (v["compress_s"] * 1e3, v["decompress_s"] * 1e3), {a: round(b * 1e3, 2) for a, b in v.get("compress_stages_s", {}).items()})
print("streams_identical", d.get("streams_identical"))
PY
