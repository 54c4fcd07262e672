cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh ab_tw1_c1 dctz_amd/lib_cut_tw1/libdctzhip.so --config c1 | sed 's/^/c1 /'
bash tools/ab_bench.sh ab_tw1_c2 dctz_amd/lib_cut_tw1/libdctzhip.so --config c2 | sed 's/^/c2 /'
python3 tools/small_bench.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('A', d['case'][:30], d.get('compress_us', d.get('compress_batch_us')), d.get('decompress_us', d.get('decompress_batch_us')))"
DCTZHIP_LIBRARY=$GRAFT_REPO_ROOT/dctz_amd/lib_cut_tw1/libdctzhip.so python3 tools/small_bench.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('B', d['case'][:30], d.get('compress_us', d.get('compress_batch_us')), d.get('decompress_us', d.get('decompress_batch_us')))"
