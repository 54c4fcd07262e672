#!/bin/bash
# Raw L2 <-> fabric request counters of the one-launch kernels on C2 (VERDICT r4 #6b): what is behind FETCH_SIZE / WRITE_SIZE.
#   bash tools/r05_c2_traffic.sh TAG
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B="python3 bench.py --config c2 --no-cpu-baseline --no-entropy-stage --steps 20 --warmup 5"
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/t1 -o t1 -- $B > /dev/null 2> $O/t1.err
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/t2 -o t2 -- $B > /dev/null 2> $O/t2.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/t3 -o t3 -- $B > /dev/null 2> $O/t3.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/t4 -o t4 -- $B > /dev/null 2> $O/t4.err
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_HIT_sum TCC_MISS_sum -d $O/t5 -o t5 -- $B > /dev/null 2> $O/t5.err
python3 tools/pmc_summary.py $O/t1 $O/t2 $O/t3 $O/t4 $O/t5 2>&1 | grep -E "^==|_one" > $O/c2_raw_traffic.txt
cat $O/c2_raw_traffic.txt
