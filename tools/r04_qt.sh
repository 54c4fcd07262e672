#!/bin/bash
# QT after a change of its kernels: the QT parity tests, then the C3 line (512^3 fp64 QT) and C2 in QT mode
set -u
TAG=${1:-r04qt}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "qt or QT or 1-" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage --config c3 > $O/bench_c3_$i.json 2> $O/bench_c3.err || exit 1
python3 -c "
import json,sys
d=json.loads(open('$O/bench_c3_$i.json').read().strip().splitlines()[-1])
print('c3', d['ms_per_step'], {k:round(v['ms'],4) for k,v in d['kernels'].items() if isinstance(v,dict)}, d['kernels'].get('compress_tail_ms'))"
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage --config c2 --mode qt > $O/bench_c2_qt.json 2> $O/bench_c2.err
python3 -c "
import json,sys
d=json.loads(open('$O/bench_c2_qt.json').read().strip().splitlines()[-1])
print('c2 qt', d['ms_per_step'])"
