export TMPDIR=/tmp
set -o pipefail
timeout -k 10 120 python3 tools/pt_scene_bench.py cbox 64 4 0 > gpurun_out/dbg.log 2>&1 || { grep -v "^  File" gpurun_out/dbg.log | head; exit 1; }
timeout -k 10 500 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu > gpurun_out/t_sk.log 2>&1 || { tail -40 gpurun_out/t_sk.log; exit 1; }
tail -2 gpurun_out/t_sk.log
SRT_DEBUG=1 python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-raster --no-cfg5 2>gpurun_out/b.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['dead_ray_elision']['ms_per_step'])"
grep pt_wave_kernel gpurun_out/b.err | sort | uniq
rm -rf gpurun_out/q; mkdir -p gpurun_out/q; cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q/f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-raster --no-elision --no-cpu-baseline --no-cfg5 --no-overlap > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q/w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-raster --no-elision --no-cpu-baseline --no-cfg5 --no-overlap > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for d in ("f", "w"):
    for path in glob.glob(f"gpurun_out/q/{d}/**/*counter_collection.csv", recursive=True):
        acc = {}
        for row in csv.DictReader(open(path)):
            if "pt_wave_kernel" in row["Kernel_Name"]: acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in acc.items(): print(k, "KB per launch", sum(v) / len(v), "n", len(v))
PY
