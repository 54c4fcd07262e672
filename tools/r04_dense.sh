#!/bin/bash
# round 4: the exception-dense 512^3 cases (k_compact_ac is the kernel under work) and the QT headline
set -o pipefail
D=gpurun_out/${1:-r04f}
mkdir -p $D
for eb in 1e-3 1e-4 1e-5; do
  python3 bench.py --no-cpu-baseline --no-entropy-stage --eb $eb --steps 50 --warmup 10 > $D/bench_f64_ec_$eb.json 2> $D/bench_f64_ec_$eb.err
done
python3 bench.py --config c3 --no-cpu-baseline --no-entropy-stage --steps 50 --warmup 10 > $D/bench_c3.json 2> $D/bench_c3.err
python3 - <<PY
import json
for nm in ("f64_ec_1e-3", "f64_ec_1e-4", "f64_ec_1e-5", "c3"):
    try:
        d = json.loads(open("$D/bench_%s.json" % nm).read().strip().splitlines()[-1])
        k = d["kernels"]
        print(nm, "step %.4f unsettled %.4f" % (d["ms_per_step"], d["unsettled_ms_per_step"]), "k_compress %.4f (%.3f)" % (k["k_compress"]["ms"], k["k_compress"]["frac"]),
              "tail %.4f" % k["compress_tail_ms"], "count %.4f" % k["decompress_count_scan_ms"], "k_decompress %.4f (%.3f)" % (k["k_decompress"]["ms"], k["k_decompress"]["frac"]), "p %.3f" % d["config"]["exception_fraction"])
    except Exception as e:
        print(nm, "no line:", e)
PY
