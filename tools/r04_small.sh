#!/bin/bash
# round 4: the mid-size configs (C1, C2, C5) through the one-launch kernels and through the chain, same box
set -o pipefail
D=gpurun_out/${1:-r04a}
mkdir -p $D
for c in c1 c2; do
  python3 bench.py --config $c --no-cpu-baseline --no-entropy-stage > $D/bench_$c.json 2> $D/bench_$c.err
  DCTZHIP_ONE=0 python3 bench.py --config $c --no-cpu-baseline --no-entropy-stage > $D/bench_${c}_chain.json 2> $D/bench_${c}_chain.err
done
python3 - <<PY
import json
for c in ("c1", "c2"):
    for v in ("", "_chain"):
        try:
            d = json.loads(open("$D/bench_%s%s.json" % (c, v)).read().strip().splitlines()[-1])
            print(c + v, "ms/step", d["ms_per_step"], "value", d["value"], "roofline", d.get("roofline"))
        except Exception as e:
            print(c + v, "no line:", e)
PY
