export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py --steps 8 --warmup 2 > gpurun_out/bench_r02_a.json 2> gpurun_out/bench_r02_a.err || tail -20 gpurun_out/bench_r02_a.err
cat gpurun_out/bench_r02_a.json
