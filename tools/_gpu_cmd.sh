export TMPDIR=/tmp
export SRT_WARM_FULL=1 SRT_STREAM_TIMES=1
python3 tools/pt_scene_bench.py blob7 1024 64 7,2 2>&1 | grep -E "mode |per-kernel"
