#!/bin/bash
# Copies the summaries of a tools/profile_round3.sh run (gpurun_out/<dir>) into profiles/ under the round's names.
#   bash tools/collect_profiles.sh r03 [r03]     (directory under gpurun_out/, prefix in profiles/)
set -eu
D=gpurun_out/${1:-r03}
P=profiles/${2:-r03}
last() { tail -n 1 "$1" > "$2"; }
last $D/bench.json ${P}_bench.json
last $D/bench_driver_form.json ${P}_bench_driver_form.json
last $D/bench_driver_form_unsettled.json ${P}_bench_driver_form_unsettled.json
last $D/warm_probe.json ${P}_warm_probe.json
for c in c1 c2 c3 c5 c5_qt; do last $D/bench_$c.json ${P}_bench_$c.json; done
for f in f32_ec_1e-3 f32_ec_1e-4 f32_ec_1e-5 f32_qt f64_ec_1e-4 f64_ec_1e-5; do last $D/bench_$f.json ${P}_bench_$f.json; done
cp $D/kernel_stats.csv ${P}_rocprof_kernel_stats.csv
for c in c1 c2 c5 f64_ec_1e-4 f64_ec_1e-5 f64_qt; do cp $D/kernel_stats_$c.csv ${P}_rocprof_kernel_stats_$c.csv; done
cp $D/pmc.txt ${P}_pmc.txt
cp $D/pmc_traffic.txt ${P}_pmc_traffic.txt
cp $D/pmc_traffic.json ${P}_pmc_traffic.json
cp $D/small_calls.json ${P}_small_calls.json
cp $D/small_calls_kernel_stats.csv ${P}_small_calls_kernel_stats.csv
ls -la profiles | grep "${2:-r03}_" | wc -l
