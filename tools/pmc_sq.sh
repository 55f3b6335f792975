#!/bin/bash
# SQ counter passes over the path-tracer bench (two passes of <= 8 counters); outputs under gpurun_out/pmc_<tag>/
set -e
tag=${1:-sq}
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d "$out/a" -- python3 "$root/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-raster --no-overlap --no-elision > "$out/a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d "$out/b" -- python3 "$root/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-raster --no-overlap --no-elision > "$out/b.log" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
for sub in ("a", "b"):
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(path)):
            if "pt_wave_kernel" in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k:24s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
