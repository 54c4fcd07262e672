#!/bin/bash
# interleaved decode forced for every type / mode (DCTZHIP_DEC_IL=2) against the default (1: fp64 EC only)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04il
for rep in 1 2 3; do
  for il in 2 1; do
    for args in "--dtype f32" "--config c3" "--dtype f32 --mode qt"; do
      DCTZHIP_DEC_IL=$il timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage $args > gpurun_out/r04il/b.json 2>/dev/null || exit 1
      python3 -c "
import json
d=json.loads(open('gpurun_out/r04il/b.json').read().strip().splitlines()[-1])
print('il=$il', '$args'.ljust(22), 'step %.4f' % d['ms_per_step'], 'k_decompress %.4f' % d['kernels']['k_decompress']['ms'], 'k_compress %.4f' % d['kernels']['k_compress']['ms'])"
    done
  done
done
