#!/bin/bash
# what the part's clocks and power do under the headline workload: bench.py for ~4 s in the background, rocm-smi polled beside it
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 bench.py --no-cpu-baseline --no-entropy-stage --steps 6000 --warmup 20 > gpurun_out/clock_watch_bench.json 2> /dev/null &
BP=$!
sleep 12      # (import torch + the generator)
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk\|power" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.3
done
wait $BP
python3 -c "
import json
d=json.loads(open('gpurun_out/clock_watch_bench.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items() if isinstance(v,dict)})"
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | tr -s ' ' | tr '\n' ';'; echo " (idle)"
