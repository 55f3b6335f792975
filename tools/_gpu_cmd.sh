export TMPDIR=/tmp
set -o pipefail
timeout -k 10 500 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu > gpurun_out/t_sk.log 2>&1 || { tail -40 gpurun_out/t_sk.log; exit 1; }
tail -2 gpurun_out/t_sk.log
for sl in 1572864 2097152 3145728; do echo "slots $sl"; SRT_STREAM_SLOTS=$sl python3 bench.py --scene cfg5 --steps 4 --no-cpu-baseline --no-raster --no-elision 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['image_sha256_16'], d['roofline'].get('stream_kernels_ms'))"; done
