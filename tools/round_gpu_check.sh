set -o pipefail
mkdir -p gpurun_out/r02h
python -m pytest tests -m gpu -x -q > gpurun_out/r02h/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02h/pytest_gpu.log; tail -3 gpurun_out/r02h/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02h/smoke.log 2>&1; tail -1 gpurun_out/r02h/smoke.log
bash tools/profile_round.sh r02h > gpurun_out/r02h/profile_round.log 2>&1
python3 bench.py --no-cpu-baseline --dtype f32 --eb 1e-4 > gpurun_out/r02h/bench_f32_ec_1e-4.json 2> gpurun_out/r02h/bench_f32.err
python3 bench.py --no-cpu-baseline --mode qt > gpurun_out/r02h/bench_f64_qt.json 2> gpurun_out/r02h/bench_qt.err
python3 bench.py --no-cpu-baseline --dtype f32 --mode qt --eb 1e-4 > gpurun_out/r02h/bench_f32_qt.json 2> gpurun_out/r02h/bench_f32qt.err
python3 bench.py --no-cpu-baseline --no-speculation > gpurun_out/r02h/bench_nospec.json 2> gpurun_out/r02h/bench_nospec.err
cat gpurun_out/r02h/bench.json
