#!/bin/bash
# Round profile set, run ON THE GPU BOX from the repo root:  bash tools/profile_round.sh r02
# 1. rocprofv3 --kernel-trace --stats of the default bench command        -> gpurun_out/<tag>/kernel_stats.csv
# 2. separate --pmc passes (never combined with trace domains other than kernel-trace): instruction mix / wave
#    time split / LDS, and the two HBM traffic counters (FETCH_SIZE, WRITE_SIZE) each in a pass of its own
# 3. the plain bench line                                                 -> gpurun_out/<tag>/bench.json
# The traffic record is placed where bench.py looks for it before the plain bench line is taken; the summaries
# (not the raw databases) are then copied into profiles/ by hand.
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
BENCH="python3 bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $BENCH > $O/kt_bench.json 2> $O/kt.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p1 -o p1 -- $BENCH --steps 5 --warmup 2 > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/p2 -o p2 -- $BENCH --steps 5 --warmup 2 > /dev/null 2> $O/p2.err
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 -d $O/p3 -o p3 -- $BENCH --steps 5 --warmup 2 > /dev/null 2> $O/p3.err
rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf -- $BENCH --steps 5 --warmup 2 > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- $BENCH --steps 5 --warmup 2 > /dev/null 2> $O/pw.err
python3 tools/pmc_summary.py $O/kt > $O/kernel_stats.csv 2>&1
python3 tools/pmc_summary.py $O/p1 $O/p2 $O/p3 > $O/pmc.txt 2>&1
python3 tools/pmc_summary.py $O/pf $O/pw > $O/pmc_traffic.txt 2>&1
python3 tools/pmc_traffic_json.py $O/pmc_traffic.txt f64_512_ec_0.001 > $O/pmc_traffic.json && cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err
cat $O/kernel_stats.csv; cat $O/pmc_traffic.txt | cut -c1-300
