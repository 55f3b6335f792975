#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline objects refer to.  Run on the GPU box from the repo root:
#   bash tools/collect_profiles.sh <tag> [scene]      scene = cbox (default) | cfg5; outputs under gpurun_out/prof_<tag>/
# Separate passes (kernel trace; FETCH_SIZE; WRITE_SIZE + L2 hit/miss; two SQ counter passes) as MI355X_MICROARCH.md prescribes:
# counters are never combined with a trace.  tools/make_traffic.py turns the CSVs into profiles/<tag>_*.
set -e
tag=${1:-r02_final}
scene=${2:-cbox}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
if [ "$scene" = raster ]; then
  # the rasterizer on BASELINE configs[1] and on the stress frame: kernel trace, two SQ counter passes, FETCH_SIZE, WRITE_SIZE (+ L2 hit / miss)
  # over tools/raster_bench.py (full frames of the resident stream)
  cd /tmp
  for wl in cfg2 stress; do
    if [ $wl = cfg2 ]; then fx=raster_cfg2_test3_1024_ss4.npz; n=100; m=20; else fx=stress_degenerate2_1024_ss4.npz; n=20; m=5; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${wl}_stats" -- python3 "$root/tools/raster_bench.py" $fx $n > "$out/${wl}_stats.log" 2>&1
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d "$out/${wl}_a" -- python3 "$root/tools/raster_bench.py" $fx $m > "$out/${wl}_a.log" 2>&1
    rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU --output-format csv -d "$out/${wl}_b" -- python3 "$root/tools/raster_bench.py" $fx $m > "$out/${wl}_b.log" 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${wl}_fetch" -- python3 "$root/tools/raster_bench.py" $fx $m > "$out/${wl}_fetch.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/${wl}_write" -- python3 "$root/tools/raster_bench.py" $fx $m > "$out/${wl}_write.log" 2>&1
    echo "$wl passes done"
  done
  cd "$root"
  python3 tools/make_raster_profile.py "$out" "$tag"
  exit 0
fi
common="--scene $scene --no-cpu-baseline --no-cfg5 --no-overlap --no-dropin --no-golden-check"
cd /tmp
if [ "$scene" = cbox ]; then statsopt="--steps 8 --warmup 1"; else statsopt="--steps 2 --warmup 1 --no-raster --no-elision"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" $statsopt $common > "$out/stats.log" 2>&1
echo "stats pass done"
pmcopt="--steps 2 --warmup 0 --no-raster --no-elision $common"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 "$root/bench.py" $pmcopt > "$out/fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/write" -- python3 "$root/bench.py" $pmcopt > "$out/write.log" 2>&1
echo "write pass done"
sq=$root/gpurun_out/pmc_$tag
rm -rf "$sq"; mkdir -p "$sq"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d "$sq/a" -- python3 "$root/bench.py" $pmcopt > "$sq/a.log" 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d "$sq/b" -- python3 "$root/bench.py" $pmcopt > "$sq/b.log" 2>&1
echo "sq passes done"
cd "$root"
python3 tools/make_traffic.py "$out" "$tag" "$sq" "$scene" 4
