#!/bin/bash
# entropy stage under its chunk size (DCTZ_DFL_THREADS builds): time and stream sizes
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in dctz_amd/lib dctz_amd/lib_cut_dfl64 dctz_amd/lib_cut_dfl256; do
  [ -f $lib/libdctzhip.so ] || continue
  echo "== $lib"
  DCTZHIP_LIBRARY=$PWD/$lib/libdctzhip.so python3 tools/deflate_bench.py --reps 10 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: v for k, v in d.items() if k not in ('inflate_ms',)})"
done
