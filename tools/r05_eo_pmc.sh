#!/bin/bash
# SQ counters of k_compress (DCTZHIP_EO=0) and k_compress_eo (=1) on one box, separate --pmc passes.
#   bash tools/r05_eo_pmc.sh TAG [bench args ...]
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
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p1_$eo -o p1 -- $B "$@" > /dev/null 2> $O/p1_$eo.err
  timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/p2_$eo -o p2 -- $B "$@" > /dev/null 2> $O/p2_$eo.err
  timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_FLAT GRBM_GUI_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 -d $O/p3_$eo -o p3 -- $B "$@" > /dev/null 2> $O/p3_$eo.err
  python3 tools/pmc_summary.py $O/p1_$eo $O/p2_$eo $O/p3_$eo 2>&1 | grep -E "^==|k_compress" > $O/pmc_eo$eo.txt
done
cat $O/pmc_eo0.txt $O/pmc_eo1.txt
