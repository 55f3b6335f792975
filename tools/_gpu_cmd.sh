export TMPDIR=/tmp
set -o pipefail
timeout -k 10 500 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu > gpurun_out/t_sk.log 2>&1 || { tail -40 gpurun_out/t_sk.log; exit 1; }
tail -2 gpurun_out/t_sk.log
export SRT_WARM_FULL=1 SRT_STREAM_TIMES=1
python3 tools/pt_scene_bench.py blob7 1024 64 7,6 2>&1 | grep -E "mode |per-kernel"
