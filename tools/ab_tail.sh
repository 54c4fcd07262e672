# compress tail (k_compact_ac) A/B between builds: bash tools/ab_tail.sh lib lib_cut_x ...
for lib in "$@"; do
  for dt in f64 f32; do for eb in 1e-4 1e-5 1e-6; do
    DCTZHIP_LIBRARY=dctz_amd/$lib/libdctzhip.so python3 tools/ab_bench.py --variants "fd=2" --rounds 9 --eb $eb --dtype $dt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib'.ljust(16), '$dt', '$eb', d['compress_ms'], d['compress_tail_ms'], d['decompress_ms'], d['p'])"
  done; done
done
