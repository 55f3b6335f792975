export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu -k "particle or tonemap" > gpurun_out/t_part.log 2>&1 || { tail -40 gpurun_out/t_part.log; exit 1; }
tail -2 gpurun_out/t_part.log
