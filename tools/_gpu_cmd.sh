export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu > gpurun_out/t_stream5.log 2>&1 || { tail -30 gpurun_out/t_stream5.log; exit 1; }
tail -2 gpurun_out/t_stream5.log
python3 tools/pt_scene_bench.py blob7 1024 64 2,7 2>&1 | grep -E "mode"
SRT_ELIDE=1 python3 tools/pt_scene_bench.py blob7 1024 64 7 2>&1 | grep -E "mode"
python3 tools/pt_scene_bench.py cbox_particles 512 16 4,6 2>&1 | grep -E "mode"
SRT_ELIDE=1 python3 tools/pt_scene_bench.py cbox 1024 64 2 2>&1 | grep -E "mode"
