export TMPDIR=/tmp
set -o pipefail
timeout -k 10 500 python3 -m pytest tests/test_pt_gpu.py -x -q -m gpu > gpurun_out/t_sk.log 2>&1 || { tail -40 gpurun_out/t_sk.log; exit 1; }
tail -2 gpurun_out/t_sk.log
SRT_WARM_FULL=1 SRT_STREAM_TIMES=1 python3 tools/pt_scene_bench.py blob7 1024 64 7,6 2>&1 | grep -E "mode |per-kernel"
bash tools/pmc_run.sh logic_c "pt_wave_kernel" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA" -- tools/pt_scene_bench.py blob7 1024 64 7 > gpurun_out/pmc_logic_c.txt 2>&1
cat gpurun_out/pmc_logic_c.txt
