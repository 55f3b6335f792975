export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu > gpurun_out/t_sk.log 2>&1 || { tail -40 gpurun_out/t_sk.log; exit 1; }
tail -2 gpurun_out/t_sk.log
SRT_CAST_STATS=1 python3 tools/pt_scene_bench.py blob7 1024 16 7 2>&1 | grep -E "cast|mode"
python3 tools/pt_scene_bench.py blob7 1024 64 6,7 2>&1 | grep -E "mode"
python3 tools/pt_scene_bench.py cbox_particles 512 16 6 2>&1 | grep -E "mode"
