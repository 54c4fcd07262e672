cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in dctz_amd/lib dctz_amd/lib_cut_nobins dctz_amd/lib_cut_binnt; do
  DCTZHIP_LIBRARY=$PWD/$lib/libdctzhip.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-entropy-stage > gpurun_out/ab2.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/ab2.json').read().strip().splitlines()[-1])
print('$lib'.ljust(28), 'step %.4f' % d['ms_per_step'], 'k_compress %.4f' % d['kernels']['k_compress']['ms'], 'k_decompress %.4f' % d['kernels']['k_decompress']['ms'], 'count %.4f' % d['kernels']['decompress_count_scan_ms'])"
done
done
