export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_raster_gpu.py -x -q -m gpu > gpurun_out/t_raster.log 2>&1 || { tail -40 gpurun_out/t_raster.log; exit 1; }
tail -2 gpurun_out/t_raster.log
python3 tools/raster_bench.py raster_cfg2_test3_1024_ss4.npz 100
python3 tools/raster_bench.py stress_degenerate2_1024_ss4.npz 10
root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_raster -- python3 $root/tools/raster_bench.py raster_cfg2_test3_1024_ss4.npz 100 > $root/gpurun_out/prof_raster.log 2>&1
cd $root
f=$(ls gpurun_out/prof_raster/*/*kernel_stats.csv | head -1)
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    print(r["Name"][:70], r["Calls"], round(float(r["TotalDurationNs"])/1e6,3), "ms avg", round(float(r["AverageNs"])/1e3,2),"us")
PY
