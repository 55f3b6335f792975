#!/bin/bash
# kernel durations of tools/hit_bench.py (arguments passed through); output under gpurun_out/hit_prof/
root=$(pwd); out=$root/gpurun_out/hit_prof; rm -rf "$out"; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/s" -- python3 "$root/tools/hit_bench.py" "$@" > "$out/run.log" 2>&1
cat "$out/run.log" | tail -5
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, os
for path in glob.glob(os.path.join(sys.argv[1], "s", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        print(row["Name"][:60], "calls", row["Calls"], "avg ms", float(row["AverageNs"]) / 1e6, "max ms", float(row["MaxNs"]) / 1e6)
PY
