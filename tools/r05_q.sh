cd $GRAFT_REPO_ROOT
for eb in 1e-3 1e-4 1e-5; do for v in "0 0" "1 0" "1 1"; do set -- $v
for rep in 1 2; do
DCTZHIP_EO=$1 DCTZHIP_EO_DIRECT=$2 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage --steps 20 --warmup 5 --eb $eb 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eb $eb eo $1 direct $2 step', round(d['ms_per_step'],4), 'unsettled', round(d.get('unsettled_ms_per_step',0),4), {a:round(b['ms'],4) for a,b in d['kernels'].items() if isinstance(b,dict) and 'ms' in b})"
done; done; done
