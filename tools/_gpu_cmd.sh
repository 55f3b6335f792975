export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu > gpurun_out/t_sk.log 2>&1 || { tail -40 gpurun_out/t_sk.log; exit 1; }
tail -2 gpurun_out/t_sk.log
for k in 32 13 10 8; do echo "lds frames $k"; SRT_DEBUG=1 SRT_CAST_LDS_FRAMES=$k python3 tools/pt_scene_bench.py blob7 1024 64 7 2>&1 | grep -E "mode |cast_kernel"; done
for k in 32 13; do echo "lds frames $k"; SRT_DEBUG=1 SRT_CAST_LDS_FRAMES=$k python3 tools/pt_scene_bench.py blob7 1024 64 6 2>&1 | grep -E "mode |cast_kernel"; done
for l in 4 12 16; do echo "leaf_min $l"; SRT_CAST_LEAF=$l python3 tools/pt_scene_bench.py blob7 1024 64 7 2>&1 | grep -E "mode "; done
