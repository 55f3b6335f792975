#!/bin/bash
# usage: tools/pmc_run.sh <tag> <kernel-substring> "<counters pass 1>" ["<counters pass 2>" ...] -- python-script args...
# Runs one rocprofv3 --pmc pass per counter list over `python3 <script> <args>` and prints the per-launch mean
# of every counter for kernels whose name contains the substring.
tag=$1; kern=$2; shift 2
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done
shift
root=$(pwd); out=$root/gpurun_out/pmc_$tag; mkdir -p "$out"; export TMPDIR=/tmp
script=$root/$1; shift
cd /tmp
i=0
for p in "${passes[@]}"; do
  rocprofv3 --pmc $p --output-format csv -d "$out/p$i" -- python3 "$script" "$@" > "$out/p$i.log" 2>&1 || tail -5 "$out/p$i.log"
  i=$((i+1))
done
cd "$root"
python3 - "$out" "$kern" <<'PY'
import csv, glob, sys, os
out, kern = sys.argv[1], sys.argv[2]
for path in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    acc = {}
    for row in csv.DictReader(open(path)):
        if kern in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:28s} mean {sum(v)/len(v):.4g}  sum {sum(v):.4g} (n={len(v)})")
PY
