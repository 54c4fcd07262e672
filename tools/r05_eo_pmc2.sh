#!/bin/bash
# instruction-class counters of k_compress (DCTZHIP_EO=0) and k_compress_eo (=1):  bash tools/r05_eo_pmc2.sh TAG
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B="python3 bench.py --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2"
for eo in 0 1; do
  export DCTZHIP_EO=$eo
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH -d $O/q1_$eo -o q1 -- $B "$@" > /dev/null 2> $O/q1_$eo.err
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/q2_$eo -o q2 -- $B "$@" > /dev/null 2> $O/q2_$eo.err
  python3 tools/pmc_summary.py $O/q1_$eo $O/q2_$eo 2>&1 | grep -E "k_compress" | grep -v "0, true>" 
done
