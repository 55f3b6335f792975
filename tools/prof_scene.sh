#!/bin/bash
# rocprofv3 kernel statistics of tools/pt_scene_bench.py on a named scene: tools/prof_scene.sh <tag> <scene> <size> <spp> <modes>
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/prof_$tag; rm -rf "$out"; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
SRT_ELIDE=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/s" -- python3 "$root/tools/pt_scene_bench.py" "$@" > "$out/run.log" 2>&1
grep "^mode\|^build" "$out/run.log"
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, os
for path in glob.glob(os.path.join(sys.argv[1], "s", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if float(row["Percentage"]) > 0.5: print(row["Name"][:70], "calls", row["Calls"], "avg ms", round(float(row["AverageNs"]) / 1e6, 3))
PY
