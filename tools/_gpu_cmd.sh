export TMPDIR=/tmp
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo rc=$?; tail -2 gpurun_out/bench_default.err; wc -c gpurun_out/bench_default.json
