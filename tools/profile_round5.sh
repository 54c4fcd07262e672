#!/bin/bash
# Round-5 profile set, run ON THE GPU BOX from the repo root:  bash tools/profile_round5.sh r05 [ABC]
#  A. the default bench line as the driver runs it (20 steps after 5), the long form; rocprofv3 --kernel-trace --stats of the
#     default command -> kernel_stats.csv; separate --pmc passes (never combined with other trace domains): FETCH_SIZE /
#     WRITE_SIZE each in a pass of its own -> pmc_traffic.json (keyed by the FULL kernel name and the hash of the kernel
#     sources), for the headline and for C2; SQ counters of the headline
#  B. every BASELINE configuration at its own size: bench.py --config c1 | c2 | c3 | c5 (+ c5 in QT mode); kernel stats c1, c2,
#     c3, c5; raw L2 <-> fabric request counters of the one-launch kernels on C2 (tools/r05_c2_traffic.sh)
#  C. k_compress_eo (DCTZHIP_EO=1: a block over two lanes) beside k_compress: alternating benches at eb 1e-3 / 1e-4 / 1e-5 with the
#     lists and with single-pass placement (DCTZHIP_EO_DIRECT=1), kernel stats, SQ counters and instruction classes of both
#     (tools/r05_eo_ab.sh, r05_eo_pmc.sh, r05_eo_pmc2.sh); the exception-density sweep of the default path; the drop-in end to end
# Summaries (not the raw databases) are copied into profiles/ afterwards (tools/collect_profiles5.sh).
set -u
TAG=${1:-r05}
PART=${2:-ABC}          # a gpurun call is 20 minutes at most: the set is taken in parts
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B="python3 bench.py"
if [[ $PART == *A* ]]; then
timeout -k 10 300 $B --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
echo "driver form done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $B --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.err
python3 tools/pmc_summary.py $O/kt > $O/kernel_stats.csv 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/pf.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/pw.err
python3 tools/pmc_summary.py $O/pf $O/pw > $O/pmc_traffic.txt 2>&1
python3 tools/pmc_traffic_json.py $O/pmc_traffic.txt c4_f64_512_ec_0.001 > $O/pmc_traffic.json
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pf_c2 -o pf -- $B --config c2 --no-cpu-baseline --no-entropy-stage --steps 20 --warmup 5 > /dev/null 2> $O/pf_c2.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pw_c2 -o pw -- $B --config c2 --no-cpu-baseline --no-entropy-stage --steps 20 --warmup 5 > /dev/null 2> $O/pw_c2.err
python3 tools/pmc_summary.py $O/pf_c2 $O/pw_c2 > $O/pmc_traffic_c2.txt 2>&1
python3 tools/pmc_traffic_json.py $O/pmc_traffic_c2.txt c2_f32_512_ec_0.0001 $O/pmc_traffic.json > $O/pmc_traffic2.json && mv $O/pmc_traffic2.json $O/pmc_traffic.json
echo "headline + c2 pmc done (copy $O/pmc_traffic.json to profiles/pmc_traffic.json and profiles/${TAG}_pmc_traffic.json afterwards)"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p1 -o p1 -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/p1.err
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/p2 -o p2 -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/p2.err
python3 tools/pmc_summary.py $O/p1 $O/p2 > $O/pmc.txt 2>&1
timeout -k 10 400 $B > $O/bench.json 2> $O/bench.err
echo "long form done"
fi
if [[ $PART == *B* ]]; then
for c in c1 c2 c3 c5; do
  timeout -k 10 400 $B --config $c > $O/bench_$c.json 2> $O/bench_$c.err
done
timeout -k 10 400 $B --config c5 --mode qt --no-cpu-baseline > $O/bench_c5_qt.json 2> $O/bench_c5_qt.err
echo "configs done"
for c in c1 c2 c3 c5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_$c -o kt -- $B --config $c --no-cpu-baseline --no-entropy-stage --steps 50 --warmup 10 > /dev/null 2> $O/kt_$c.err
  python3 tools/pmc_summary.py $O/kt_$c > $O/kernel_stats_$c.csv 2>&1
done
echo "kernel traces done"
bash tools/r05_c2_traffic.sh $TAG > /dev/null 2>&1
echo "c2 raw traffic done"
fi
if [[ $PART == *C* ]]; then
bash tools/r05_eo_ab.sh $TAG/eo --steps 20 --warmup 5 > $O/eo_ab.txt 2>&1
bash tools/r05_eo_pmc.sh $TAG/eo > $O/eo_pmc.txt 2>&1
bash tools/r05_eo_pmc2.sh $TAG/eo > $O/eo_pmc2.txt 2>&1
python3 tools/pmc_summary.py $O/eo/q1_1 $O/eo/q2_1 2>&1 | grep "k_compress_eo" >> $O/eo_pmc2.txt
bash tools/r05_q.sh > $O/eo_density_sweep.txt 2>&1
echo "eo done"
for eb in 1e-4 1e-5; do
  timeout -k 10 300 $B --no-cpu-baseline --no-entropy-stage --eb $eb > $O/bench_f64_ec_$eb.json 2> $O/bench_f64_$eb.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_f64_$eb -o kt -- $B --no-cpu-baseline --no-entropy-stage --eb $eb --steps 50 --warmup 10 > /dev/null 2> $O/kt_f64_$eb.err
  python3 tools/pmc_summary.py $O/kt_f64_$eb > $O/kernel_stats_f64_ec_$eb.csv 2>&1
done
timeout -k 10 600 python3 tools/e2e_bench.py --skip-reference-tail --threads 16 > $O/e2e_dropin.json 2> $O/e2e_dropin.err
echo "all done"
fi
head -c 700 $O/bench_driver_form.json 2>/dev/null; echo
