export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest tests/test_dropin_gpu.py -x -q -m gpu > gpurun_out/t_dropin.log 2>&1 || { tail -40 gpurun_out/t_dropin.log; exit 1; }
tail -3 gpurun_out/t_dropin.log
