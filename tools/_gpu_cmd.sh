export TMPDIR=/tmp
export SRT_WARM_FULL=1
for o in 20 24 32 48; do echo "object_min $o"; SRT_CAST_OBJECT=$o python3 tools/pt_scene_bench.py blob7 1024 64 7 2>&1 | grep -E "mode "; done
for o in 24 32; do echo "object_min $o leaf 12"; SRT_CAST_LEAF=12 SRT_CAST_OBJECT=$o python3 tools/pt_scene_bench.py blob7 1024 64 7 2>&1 | grep -E "mode "; done
