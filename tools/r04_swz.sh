#!/bin/bash
# after a change of the tile image's swizzle: parity, the headline line, LDS counters of the two big kernels
set -u
TAG=${1:-r04swz}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage > $O/bench.json 2> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --no-entropy-stage > $O/kt_bench.json 2> $O/kt.err
python3 tools/pmc_summary.py $O/kt > $O/kernel_stats.csv 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES -d $O/p2 -o p2 -- python3 bench.py --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/p2.err
python3 tools/pmc_summary.py $O/p2 > $O/pmc_lds.txt 2>&1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage --dtype f32 > $O/bench_f32.json 2> $O/bench_f32.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage --config c2 > $O/bench_c2.json 2> $O/bench_c2.err
head -c 600 $O/bench.json; echo; grep "k_compress<\|k_decompress<" $O/kernel_stats.csv | cut -c1-120; cat $O/pmc_lds.txt | cut -c1-300
