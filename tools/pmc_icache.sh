#!/bin/bash
root=$(pwd); out=$root/gpurun_out/pmc_icache; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > "$out/avail.txt" 2>&1
grep -o -i "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*" "$out/avail.txt" | sort -u | tr '\n' ' '
echo
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d "$out/a" -- python3 "$root/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-raster > "$out/a.log" 2>&1 || tail -5 "$out/a.log"
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
for path in glob.glob(os.path.join(out, "a", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for row in csv.DictReader(open(path)):
        if "pt_wave_kernel" in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:24s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
