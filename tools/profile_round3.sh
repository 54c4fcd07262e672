#!/bin/bash
# Round-3 profile set, run ON THE GPU BOX from the repo root:  bash tools/profile_round3.sh r03
#  1. the default bench line as the driver runs it (20 steps after 5) and the long form (200 after 20)
#  2. rocprofv3 --kernel-trace --stats of the default bench command            -> kernel_stats.csv
#  3. separate --pmc passes (never combined with other trace domains): HBM FETCH_SIZE / WRITE_SIZE each in a pass of
#     its own -> pmc_traffic.json (the record bench.py quotes), instruction mix / LDS of the two big kernels -> pmc.txt
#  4. every BASELINE configuration at its own size: bench.py --config c1 | c2 | c3 | c5 (+ c5 in QT mode, what
#     tests/test-dctz.sh runs) with rocprof kernel stats for c1, c2, c5; the exception-density sweep (eb 1e-4, 1e-5)
#  5. tools/small_bench.py (one call per array against the batch entry points) + its kernel trace
# Summaries (not the raw databases) are copied into profiles/ by hand afterwards.
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B="python3 bench.py"
timeout -k 10 300 $B --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
timeout -k 10 300 $B --steps 20 --warmup 5 --settle-ms 0 --no-cpu-baseline --no-entropy-stage > $O/bench_driver_form_unsettled.json 2> $O/bench_driver_form_unsettled.err
python3 tools/warm_probe.py 80 > $O/warm_probe.json 2> $O/warm_probe.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $B --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.err
python3 tools/pmc_summary.py $O/kt > $O/kernel_stats.csv 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/pw.err
python3 tools/pmc_summary.py $O/pf $O/pw > $O/pmc_traffic.txt 2>&1
python3 tools/pmc_traffic_json.py $O/pmc_traffic.txt c4_f64_512_ec_0.001 > $O/pmc_traffic.json && cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p1 -o p1 -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/p2 -o p2 -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/p2.err
python3 tools/pmc_summary.py $O/p1 $O/p2 > $O/pmc.txt 2>&1
timeout -k 10 400 $B > $O/bench.json 2> $O/bench.err
for c in c1 c2 c3 c5; do
  timeout -k 10 400 $B --config $c > $O/bench_$c.json 2> $O/bench_$c.err
done
timeout -k 10 400 $B --config c5 --mode qt --no-cpu-baseline > $O/bench_c5_qt.json 2> $O/bench_c5_qt.err
for c in c1 c2 c5; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$c -o kt -- $B --config $c --no-cpu-baseline --no-entropy-stage --steps 50 --warmup 10 > /dev/null 2> $O/kt_$c.err
  python3 tools/pmc_summary.py $O/kt_$c > $O/kernel_stats_$c.csv 2>&1
done
for eb in 1e-4 1e-5; do
  timeout -k 10 300 $B --no-cpu-baseline --no-entropy-stage --eb $eb > $O/bench_f64_ec_$eb.json 2> $O/bench_f64_$eb.err
  timeout -k 10 300 $B --no-cpu-baseline --no-entropy-stage --dtype f32 --eb $eb > $O/bench_f32_ec_$eb.json 2> $O/bench_f32_$eb.err
  rocprofv3 --kernel-trace --stats -d $O/kt_f64_$eb -o kt -- $B --no-cpu-baseline --no-entropy-stage --eb $eb --steps 50 --warmup 10 > /dev/null 2> $O/kt_f64_$eb.err
  python3 tools/pmc_summary.py $O/kt_f64_$eb > $O/kernel_stats_f64_ec_$eb.csv 2>&1
done
timeout -k 10 300 $B --no-cpu-baseline --no-entropy-stage --dtype f32 > $O/bench_f32_ec_1e-3.json 2> $O/bench_f32.err
timeout -k 10 300 $B --no-cpu-baseline --no-entropy-stage --dtype f32 --mode qt --eb 1e-4 > $O/bench_f32_qt.json 2> $O/bench_f32qt.err
rocprofv3 --kernel-trace --stats -d $O/kt_qt -o kt -- $B --config c3 --no-cpu-baseline --no-entropy-stage --steps 50 --warmup 10 > /dev/null 2> $O/kt_qt.err
python3 tools/pmc_summary.py $O/kt_qt > $O/kernel_stats_f64_qt.csv 2>&1
python3 tools/small_bench.py > $O/small_calls.json 2> $O/small_calls.err
rocprofv3 --kernel-trace --stats -d $O/kt_small -o kt -- python3 tools/small_bench.py --only batch25 --rounds 200 > /dev/null 2> $O/kt_small.err
python3 tools/pmc_summary.py $O/kt_small > $O/small_calls_kernel_stats.csv 2>&1
cat $O/bench.json | head -c 600; echo; cat $O/small_calls.json
