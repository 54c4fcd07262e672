#!/bin/bash
# Round-4 profile set, run ON THE GPU BOX from the repo root:  bash tools/profile_round4.sh r04
#  1. the default bench line as the driver runs it (20 steps after 5; the line carries the unsettled figure too) and the
#     long form (200 after 20)
#  2. rocprofv3 --kernel-trace --stats of the default bench command            -> kernel_stats.csv
#  3. separate --pmc passes (never combined with other trace domains): HBM FETCH_SIZE / WRITE_SIZE each in a pass of
#     its own -> pmc_traffic.json, keyed by the hash of the kernel sources (bench.py quotes it for THIS build only)
#  4. every BASELINE configuration at its own size: bench.py --config c1 | c2 | c3 | c5 (+ c5 in QT mode), each also
#     through the chain of kernels (DCTZHIP_ONE=0) for c1 / c2; rocprof kernel stats for c1, c2; the exception-density
#     sweep (eb 1e-4, 1e-5); where the time of the one-launch kernels goes (tools/one_stamps.py)
#  5. tools/small_bench.py (one call per array against the batch entry points)
#  6. the drop-in end to end (tools/e2e_bench.py)
# Summaries (not the raw databases) are copied into profiles/ afterwards.
set -u
TAG=${1:-r04}
PART=${2:-ABC}          # a gpurun call is 20 minutes at most: the set is taken in three parts
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B="python3 bench.py"
if [[ $PART == *A* ]]; then
timeout -k 10 300 $B --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
echo "driver form done"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $B --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.err
python3 tools/pmc_summary.py $O/kt > $O/kernel_stats.csv 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/pw.err
python3 tools/pmc_summary.py $O/pf $O/pw > $O/pmc_traffic.txt 2>&1
python3 tools/pmc_traffic_json.py $O/pmc_traffic.txt c4_f64_512_ec_0.001 > $O/pmc_traffic.json
# traffic of the one-launch kernels on C2 (the mid-size headline of this round), same record
rocprofv3 --pmc FETCH_SIZE -d $O/pf_c2 -o pf -- $B --config c2 --no-cpu-baseline --no-entropy-stage --steps 20 --warmup 5 > /dev/null 2> $O/pf_c2.err
rocprofv3 --pmc WRITE_SIZE -d $O/pw_c2 -o pw -- $B --config c2 --no-cpu-baseline --no-entropy-stage --steps 20 --warmup 5 > /dev/null 2> $O/pw_c2.err
python3 tools/pmc_summary.py $O/pf_c2 $O/pw_c2 > $O/pmc_traffic_c2.txt 2>&1
python3 tools/pmc_traffic_json.py $O/pmc_traffic_c2.txt c2_f32_512_ec_0.0001 $O/pmc_traffic.json > $O/pmc_traffic2.json && mv $O/pmc_traffic2.json $O/pmc_traffic.json
echo "headline + c2 pmc done (copy $O/pmc_traffic.json to profiles/pmc_traffic.json and profiles/${TAG}_pmc_traffic.json afterwards)"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p1 -o p1 -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/p2 -o p2 -- $B --no-cpu-baseline --no-entropy-stage --steps 5 --warmup 2 > /dev/null 2> $O/p2.err
python3 tools/pmc_summary.py $O/p1 $O/p2 > $O/pmc.txt 2>&1
timeout -k 10 400 $B > $O/bench.json 2> $O/bench.err
echo "long form done"
fi
if [[ $PART == *B* ]]; then
for c in c1 c2 c3 c5; do
  timeout -k 10 400 $B --config $c > $O/bench_$c.json 2> $O/bench_$c.err
done
for c in c1 c2; do
  DCTZHIP_ONE=0 timeout -k 10 400 $B --config $c --no-cpu-baseline --no-entropy-stage > $O/bench_${c}_chain.json 2> $O/bench_${c}_chain.err
  timeout -k 10 400 $B --config $c --mode qt --no-cpu-baseline --no-entropy-stage > $O/bench_${c}_qt.json 2> $O/bench_${c}_qt.err
done
timeout -k 10 400 $B --config c5 --mode qt --no-cpu-baseline > $O/bench_c5_qt.json 2> $O/bench_c5_qt.err
echo "configs done"
for c in c1 c2 c5; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$c -o kt -- $B --config $c --no-cpu-baseline --no-entropy-stage --steps 50 --warmup 10 > /dev/null 2> $O/kt_$c.err
  python3 tools/pmc_summary.py $O/kt_$c > $O/kernel_stats_$c.csv 2>&1
done
echo "kernel traces done"
fi
if [[ $PART == *C* ]]; then
for eb in 1e-4 1e-5; do
  timeout -k 10 300 $B --no-cpu-baseline --no-entropy-stage --eb $eb > $O/bench_f64_ec_$eb.json 2> $O/bench_f64_$eb.err
  timeout -k 10 300 $B --no-cpu-baseline --no-entropy-stage --dtype f32 --eb $eb > $O/bench_f32_ec_$eb.json 2> $O/bench_f32_$eb.err
  rocprofv3 --kernel-trace --stats -d $O/kt_f64_$eb -o kt -- $B --no-cpu-baseline --no-entropy-stage --eb $eb --steps 50 --warmup 10 > /dev/null 2> $O/kt_f64_$eb.err
  python3 tools/pmc_summary.py $O/kt_f64_$eb > $O/kernel_stats_f64_ec_$eb.csv 2>&1
done
timeout -k 10 300 $B --no-cpu-baseline --no-entropy-stage --dtype f32 > $O/bench_f32_ec_1e-3.json 2> $O/bench_f32.err
rocprofv3 --kernel-trace --stats -d $O/kt_qt -o kt -- $B --config c3 --no-cpu-baseline --no-entropy-stage --steps 50 --warmup 10 > /dev/null 2> $O/kt_qt.err
python3 tools/pmc_summary.py $O/kt_qt > $O/kernel_stats_f64_qt.csv 2>&1
echo "sweeps done"
(python3 tools/one_stamps.py c1; python3 tools/one_stamps.py c2; python3 tools/one_stamps.py c1 qt; python3 tools/one_stamps.py c2 qt) 2>&1 | grep -v amdgpu.ids > $O/one_launch_stamps.txt
python3 tools/small_bench.py > $O/small_calls.json 2> $O/small_calls.err
DCTZHIP_ONE=0 python3 tools/small_bench.py > $O/small_calls_chain.json 2> $O/small_calls_chain.err
timeout -k 10 600 python3 tools/e2e_bench.py --skip-reference-tail --threads 16 > $O/e2e_dropin.json 2> $O/e2e_dropin.err
timeout -k 10 300 python -m pytest tests/test_libdctz_gpu.py -m gpu -q -s -k host_buffer_batch 2>&1 | grep "host-buffer batch" > $O/host_batch.txt
echo "all done"
fi
head -c 700 $O/bench_driver_form.json; echo
