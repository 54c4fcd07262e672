#!/bin/bash
# A/B of library builds over the exception-density sweep: tools/ab_sweep.sh <out.json> <lib> [<lib> ...]
# (kernel medians from tools/ab_bench.py per build, dtype, mode and error bound; builds alternate inside every case)
OUT=$1; shift
: > $OUT
for cfg in "f64 ec 1e-3" "f64 ec 1e-4" "f64 ec 1e-5" "f32 ec 1e-3" "f32 ec 1e-4" "f32 ec 1e-5" "f64 qt 1e-3" "f32 qt 1e-4"; do
  set -- $cfg "$@"; dt=$1; mode=$2; eb=$3; shift 3
  for lib in "$@"; do
    echo "{\"lib\": \"$lib\", \"dtype\": \"$dt\", \"mode\": \"$mode\", \"eb\": $eb}" >> $OUT
    DCTZHIP_LIBRARY=$lib python3 tools/ab_bench.py --dtype $dt --mode $mode --eb $eb --variants "fd=2" >> $OUT 2>> $OUT.err
  done
done
