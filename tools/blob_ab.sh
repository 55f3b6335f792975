#!/bin/bash
# A/B of library builds on mesh scenes: tools/blob_ab.sh "<lib suffixes>" "<scene names>"
for v in ${1:-DEFAULT}; do [ "$v" = DEFAULT ] && v=""; for sc in $2; do echo "lib$v $sc"; SRT_ELIDE=1 SRT_HIP_LIBRARY=$PWD/soft-rendering-toolsets_amd/lib$v/libsrt_hip.so python tools/pt_scene_bench.py $sc 512 16 2,4 2>&1 | grep "^mode"; done; done
