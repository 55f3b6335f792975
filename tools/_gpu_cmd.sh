export TMPDIR=/tmp
set -o pipefail
timeout -k 10 600 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu -s -k "device_bvh" > gpurun_out/t_bvh.log 2>&1 || { tail -40 gpurun_out/t_bvh.log; exit 1; }
grep -E "build_scene|passed|failed" gpurun_out/t_bvh.log
