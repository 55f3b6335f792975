export TMPDIR=/tmp
set -o pipefail
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1 || { tail -40 gpurun_out/t_all.log; exit 1; }
tail -3 gpurun_out/t_all.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
