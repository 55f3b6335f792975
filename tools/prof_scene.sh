#!/bin/bash
# usage: tools/prof_scene.sh <tag> <kernel-substring> <scene> <size> <spp> <modes>
# Kernel trace + SQ / TA / TCP / TCC counter passes (separate passes, never combined with a trace) over
# tools/pt_scene_bench.py <scene> <size> <spp> <modes>; per-launch means of the kernels whose name contains the substring.
tag=$1; kern=$2; shift 2
root=$(pwd); out=$root/gpurun_out/prof_$tag; mkdir -p "$out"; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/tools/pt_scene_bench.py" "$@" > "$out/stats.log" 2>&1 || tail -5 "$out/stats.log"
cd "$root"
bash tools/pmc_run.sh "$tag" "$kern" \
  "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
  "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS" \
  "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum TCP_GATE_EN2_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  -- tools/pt_scene_bench.py "$@" > "$out/pmc_means.txt" 2>&1
cat "$out/pmc_means.txt"
