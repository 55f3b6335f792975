export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu > gpurun_out/t_sw.log 2>&1 || { tail -40 gpurun_out/t_sw.log; exit 1; }
tail -2 gpurun_out/t_sw.log
python3 tools/pt_scene_bench.py blob7 1024 64 7 2>&1 | grep -E "mode"
python3 tools/pt_scene_bench.py blob3 1024 64 7 2>&1 | grep -E "mode"
timeout -k 10 600 python3 bench.py --steps 4 --warmup 1 --no-raster --no-cpu-baseline --no-elision 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline', d['value'], d['ms_per_step']); c=d['cfg5']; print('cfg5', c['value'], c['ms_per_step'], c['roofline']['stream_kernels_ms'])"
