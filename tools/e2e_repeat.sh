cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
DCTZ_PIPE_DEBUG=1 python3 tools/e2e_bench.py --skip-reference-tail --threads 16 > gpurun_out/e2e_$i.json 2> gpurun_out/e2e_$i.err
python3 -c "
import json
d=json.loads(open('gpurun_out/e2e_$i.json').read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,dict) and 'compress_s' in v and 'gpu' in k: print(k, round(v['compress_s']*1e3,2), round(v['decompress_s']*1e3,2))
"
grep "cpipe\] max" gpurun_out/e2e_$i.err | tail -2
done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os; print(len(os.sched_getaffinity(0)))"
