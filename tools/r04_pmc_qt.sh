cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcqt; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p1 -o p1 -- python3 bench.py --config c3 --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/p2 -o p2 -- python3 bench.py --config c3 --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/p2.err
python3 tools/pmc_summary.py $O/p1 $O/p2 > $O/pmc_qt.txt 2>&1
grep "k_decompress\|k_compress<double, 1, true, 2, 0, false" $O/pmc_qt.txt | cut -c1-400
