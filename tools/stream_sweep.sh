#!/bin/bash
# Diagnostic sweep of the streamed form's launch parameters on one scene; prints ms / Mrays/s per setting.
scene=${1:-blob7}; size=${2:-1024}; spp=${3:-16}
run() { echo "== $*"; env "$@" python3 tools/pt_scene_bench.py $scene $size $spp 6 2>&1 | grep "mode 6"; }
run SRT_CAST_FETCH=16
run SRT_CAST_FETCH=1
run SRT_CAST_FETCH=8
run SRT_CAST_FETCH=32
run SRT_CAST_FETCH=48
run SRT_CAST_INTERIOR=1
run SRT_CAST_INTERIOR=8
run SRT_CAST_INTERIOR=32
run SRT_CAST_INTERIOR=48
run SRT_STREAM_SLOTS=262144
run SRT_STREAM_SLOTS=524288
run SRT_STREAM_SLOTS=2097152
run SRT_STREAM_SLOTS=4194304
