#!/bin/bash
# short counter set for k_compress_eo only:  bash tools/r05_eo_pmc3.sh TAG
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B="python3 bench.py --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2"
export DCTZHIP_EO=1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/s1 -o s1 -- $B "$@" > /dev/null 2> $O/s1.err
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM -d $O/s2 -o s2 -- $B "$@" > /dev/null 2> $O/s2.err
python3 tools/pmc_summary.py $O/s1 $O/s2 2>&1 | grep -E "k_compress_eo"
