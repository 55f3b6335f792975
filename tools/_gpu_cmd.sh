export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu > gpurun_out/t_stream3.log 2>&1 || { tail -30 gpurun_out/t_stream3.log; exit 1; }
tail -2 gpurun_out/t_stream3.log
SRT_DEBUG=1 SRT_CAST_STATS=1 python3 tools/pt_scene_bench.py blob7 1024 16 6,7 2>&1 | grep -E "cast|mode"
python3 tools/pt_scene_bench.py blob7 1024 64 2,6,7 2>&1 | grep -E "mode"
SRT_ELIDE=1 python3 tools/pt_scene_bench.py cbox 1024 64 2,6 2>&1 | grep -E "mode"
