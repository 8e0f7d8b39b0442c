set -e
bash tools/collect_r05.sh exact cache
python3 tools/parity_sweep.py > gpurun_out/r05/parity_sweep.log 2>&1; tail -3 gpurun_out/r05/parity_sweep.log
python3 tools/fullsize_exact.py > gpurun_out/r05/fullsize_exact.log 2>&1; tail -2 gpurun_out/r05/fullsize_exact.log
