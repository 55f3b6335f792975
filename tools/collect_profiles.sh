#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline object refers to.  Run on the GPU box from the repo root:
#   bash tools/collect_profiles.sh <tag>          (outputs under gpurun_out/prof_<tag>/)
# Separate passes (kernel trace; FETCH_SIZE; WRITE_SIZE + L2 hit/miss; two SQ counter passes) as MI355X_MICROARCH.md prescribes:
# counters are never combined with a trace.  tools/make_traffic.py turns the CSVs into profiles/<tag>_*.
set -e
tag=${1:-r01_final}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" --steps 8 --warmup 1 --no-cpu-baseline --no-overlap > "$out/stats.log" 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 "$root/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-raster --no-overlap --no-elision > "$out/fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/write" -- python3 "$root/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-raster --no-overlap --no-elision > "$out/write.log" 2>&1
echo "write pass done"
cd "$root"
bash tools/pmc_sq.sh "$tag" > "$out/pmc_sq.log" 2>&1      # SQ instruction counters (two more passes), outputs under gpurun_out/pmc_<tag>/
echo "sq passes done"
python3 tools/make_traffic.py "$out" "$tag" "$root/gpurun_out/pmc_$tag"
