#!/bin/bash
# fp64 QT (config C3): staged items as two dword planes (the tree's build) against 8-byte stores (dctz_amd/lib_qtold), alternating;
# then the LDS counters of both.   bash tools/r05_qt_ab.sh TAG
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "QT or qt or 1-" > $O/pytest_qt.log 2>&1; tail -2 $O/pytest_qt.log
for rep in 1 2 3; do for v in new old; do
  L=$R/dctz_amd/lib/libdctzhip.so; [ $v = old ] && L=$R/dctz_amd/lib_qtold/libdctzhip.so
  DCTZHIP_LIBRARY=$L timeout -k 10 300 python3 bench.py --config c3 --no-cpu-baseline --no-entropy-stage --steps 20 --warmup 5 > $O/${v}_$rep.json 2> $O/${v}_$rep.err
  python3 -c "
import json; d=json.loads(open('$O/${v}_$rep.json').read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), 'settled', round(d['settled_ms_per_step'],4), {a:round(b['ms'],4) for a,b in d['kernels'].items() if isinstance(b,dict) and 'ms' in b}, 'tail', round(d['kernels']['compress_tail_ms'],4))"
done; done
for v in new old; do
  L=$R/dctz_amd/lib/libdctzhip.so; [ $v = old ] && L=$R/dctz_amd/lib_qtold/libdctzhip.so
  DCTZHIP_LIBRARY=$L timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p_$v -o p -- python3 bench.py --config c3 --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/p_$v.err
  python3 tools/pmc_summary.py $O/p_$v 2>&1 | grep -E "k_compress<" | sed "s/^/$v /"
done
