#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts in the ray-state access patterns (tools/ubench/traffic_calib.hip).
# Run on the GPU box from the repo root; prints counter x 1024 / bytes per kernel.
set -e
root=$(pwd); out=$root/gpurun_out/traffic_calib; rm -rf "$out"; mkdir -p "$out"; export TMPDIR=/tmp
hipcc -O2 --offload-arch=gfx950 -o "$out/traffic_calib" tools/ubench/traffic_calib.hip
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- "$out/traffic_calib" > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- "$out/traffic_calib" > "$out/write.log" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
lanes, n = 8 << 20, 1 << 30
bytes_of = {"rec_store": lanes * 32, "rec_load": lanes * 96, "dword_store": n * 4, "f4_stream": n * 4}
for name in ("fetch", "write"):
    for path in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0]
            if k in bytes_of:
                print(f"{row['Counter_Name']:11s} {k:12s} counter {float(row['Counter_Value']):14.0f} KB = {float(row['Counter_Value']) * 1024 / bytes_of[k]:.3f} x the bytes the kernel moves")
PY
