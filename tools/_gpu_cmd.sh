export TMPDIR=/tmp
set -o pipefail
R=$PWD/soft-rendering-toolsets_amd
timeout -k 10 600 python3 -m pytest tests/test_raster_gpu.py tests/test_dropin_gpu.py -x -q -m gpu > gpurun_out/t_r.log 2>&1 || { tail -30 gpurun_out/t_r.log; exit 1; }
tail -2 gpurun_out/t_r.log
cd /tmp
for v in lib lib_x_b8_4 lib_x_b24_8 lib_x_b32_4; do
  echo "== $v"
  SRT_HIP_LIBRARY=$R/$v/libsrt_hip.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rb_$v -- python3 $GRAFT_REPO_ROOT/tools/raster_bench.py raster_cfg2_test3_1024_ss4.npz 100 2>&1 | grep -E "ms/frame|matches"
  python3 - $GRAFT_REPO_ROOT/gpurun_out/rb_$v <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "raster" in r["Name"] and "true" not in r["Name"]: print("   ", r["Name"][:60], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2))
PY
done
