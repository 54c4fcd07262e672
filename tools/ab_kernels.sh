#!/bin/bash
# Kernel-level A/B between BUILDS of libdctzhip.so on one box: tools/ab_bench.py (HIP-event medians of k_compress /
# k_decompress) once per library, repeated.   bash tools/ab_kernels.sh <rounds> name=path [name=path ...]
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    name=${v%%=*}; path=${v#*=}
    echo -n "$name "
    DCTZHIP_LIBRARY=$PWD/$path python3 tools/ab_bench.py --variants "fd=2" --rounds 15 ${AB_ARGS:-} 2>&1 | tail -1
  done
done
