cd $GRAFT_REPO_ROOT
for eb in 1e-4 1e-5; do for eo in 0 1; do
DCTZHIP_EO=$eo timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage --steps 20 --warmup 5 --eb $eb 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eb $eb eo $eo', round(d['ms_per_step'],4), {a:round(b['ms'],4) for a,b in d['kernels'].items() if isinstance(b,dict) and 'ms' in b}, d.get('compress_tail_ms'), d.get('p'))"
done; done
