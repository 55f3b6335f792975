export TMPDIR=/tmp
set -o pipefail
timeout -k 10 500 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu > gpurun_out/t_sk.log 2>&1 || { tail -40 gpurun_out/t_sk.log; exit 1; }
tail -2 gpurun_out/t_sk.log
export SRT_WARM_FULL=1 SRT_STREAM_TIMES=1
for o in 0 128 192; do for g in 32 64 128; do echo "own $o grab $g"; SRT_CAST_OWN=$o SRT_CAST_GRAB=$g python3 tools/pt_scene_bench.py blob7 1024 64 7 2>&1 | grep -E "per-kernel"; done; done
