#!/bin/bash
# Builds of libdctzhip.so that differ in -D flags of dctz_kernels_eo.hip, run alternately on one box with DCTZHIP_EO=1.
#   (here)  bash tools/r05_eo_variants.sh build name="-DFOO=1" [name="..."]     -> dctz_amd/lib_<name>/libdctzhip.so
#   (box)   bash tools/r05_eo_variants.sh run TAG name [name ...]                (name "base" = dctz_amd/lib; "old" = DCTZHIP_EO=0)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "$1" = build ]; then
  shift
  for v in "$@"; do
    name=${v%%=*}; flags=${v#*=}
    D=$R/dctz_amd/lib_$name
    mkdir -p $D
    for f in $R/dctz_amd/lib/*.o; do b=$(basename $f); [ $b = dctz_kernels_eo.o ] || cp -u $f $D/; done
    (cd dctz_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-pass-failed -I../include $flags -c csrc/dctz_kernels_eo.hip -o $D/dctz_kernels_eo.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libdctzhip.so $D/dctz_kernels_p*.o $D/dctz_kernels.o $D/dctz_kernels_eo.o $D/dctz_kernels_one.o $D/dctz_kernels_aux.o $D/dctz_deflate.o $D/dctz_shim.o) || exit 1
    rm -f $D/dctz_kernels_p*.o $D/dctz_kernels.o $D/dctz_kernels_one.o $D/dctz_kernels_aux.o $D/dctz_deflate.o $D/dctz_shim.o $D/dct_host.o
    echo built $D
  done
  exit 0
fi
shift; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
for rep in 1 2 3; do
  for name in "$@"; do
    if [ $name = old ]; then E=0; L=$R/dctz_amd/lib/libdctzhip.so; elif [ $name = base ]; then E=1; L=$R/dctz_amd/lib/libdctzhip.so; else E=1; L=$R/dctz_amd/lib_$name/libdctzhip.so; fi
    DCTZHIP_EO=$E DCTZHIP_LIBRARY=$L timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage --steps 20 --warmup 5 ${BENCH_ARGS:-} > $O/${name}_$rep.json 2> $O/${name}_$rep.err || echo "run $name $rep failed"
  done
done
python3 - $O "$@" <<'PY'
import json, sys, glob, statistics
o = sys.argv[1]
for name in sys.argv[2:]:
    rows = []
    for f in sorted(glob.glob(f"{o}/{name}_[0-9].json")):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        rows.append((d["kernels"]["k_compress"]["ms"], d["kernels"]["k_decompress"]["ms"], d["ms_per_step"]))
    if rows:
        print(name.ljust(10), "k_compress", [round(r[0], 4) for r in rows], "median", round(statistics.median(r[0] for r in rows), 4), " step", round(statistics.median(r[2] for r in rows), 4))
PY
