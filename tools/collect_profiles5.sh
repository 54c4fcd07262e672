#!/bin/bash
# Copies the summaries of a tools/profile_round5.sh run (gpurun_out/<dir>) into profiles/ under the round's names.
#   bash tools/collect_profiles5.sh r05 [r05]     (directory under gpurun_out/, prefix in profiles/)
set -u
D=gpurun_out/${1:-r05}
P=profiles/${2:-r05}
last() { [ -s "$1" ] && tail -n 1 "$1" > "$2"; }
cpy() { [ -s "$1" ] && cp "$1" "$2"; }
last $D/bench.json ${P}_bench.json
last $D/bench_driver_form.json ${P}_bench_driver_form.json
for c in c1 c2 c3 c5 c5_qt f64_ec_1e-4 f64_ec_1e-5; do last $D/bench_$c.json ${P}_bench_$c.json; done
cpy $D/kernel_stats.csv ${P}_rocprof_kernel_stats.csv
for c in c1 c2 c3 c5 f64_ec_1e-4 f64_ec_1e-5; do cpy $D/kernel_stats_$c.csv ${P}_rocprof_kernel_stats_$c.csv; done
cpy $D/pmc.txt ${P}_pmc.txt
cpy $D/pmc_traffic.txt ${P}_pmc_traffic.txt
cpy $D/pmc_traffic_c2.txt ${P}_pmc_traffic_c2.txt
cpy $D/pmc_traffic.json ${P}_pmc_traffic.json
cpy $D/pmc_traffic.json profiles/pmc_traffic.json
cpy $D/c2_raw_traffic.txt ${P}_c2_raw_traffic.txt
cpy $D/eo_ab.txt ${P}_eo_ab.txt
cpy $D/eo_pmc.txt ${P}_eo_pmc.txt
cpy $D/eo_pmc2.txt ${P}_eo_pmc_instruction_classes.txt
cpy $D/eo_density_sweep.txt ${P}_eo_density_sweep.txt
cpy $D/eo/kernel_stats_eo.csv ${P}_rocprof_kernel_stats_eo.csv
cpy $D/e2e_dropin.json ${P}_e2e_dropin.json
ls -la profiles | grep "${2:-r05}_" | wc -l
