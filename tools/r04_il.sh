#!/bin/bash
# tile-interleaved k_decompress (DCTZHIP_DEC_IL=1, the default) against contiguous ranges (=0): parity, then the headline and
# the exception-dense line, alternating on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04il
timeout -k 10 900 python -m pytest tests -m gpu -x -q ${IL_TESTS:-} > gpurun_out/r04il/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r04il/pytest.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do
  for il in 1 0; do
    for args in "" "--eb 1e-5" "--dtype f32" "--config c3"; do
      DCTZHIP_DEC_IL=$il timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage $args > gpurun_out/r04il/b.json 2>/dev/null || exit 1
      python3 -c "
import json
d=json.loads(open('gpurun_out/r04il/b.json').read().strip().splitlines()[-1])
print('il=$il', '$args'.ljust(12), 'step %.4f' % d['ms_per_step'], 'k_decompress %.4f' % d['kernels']['k_decompress']['ms'], 'k_compress %.4f' % d['kernels']['k_compress']['ms'])"
    done
  done
done
