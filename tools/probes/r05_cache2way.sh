set -e
mkdir -p gpurun_out/r05
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_selfplay_exact.py tests/test_gpu_multirank.py -m gpu -x -q -k "cache" 2>&1 | tail -3
timeout -k 10 500 python3 tools/leg_sweep.py cache 8192:2::3072::22 8192:2::3072::23 8192:2::3072::24 8192:2::3072::22 8192:2::3072::24 > gpurun_out/r05/sweep_cache_2way.txt 2> gpurun_out/r05/sweep_cache_2way.err
cut -c1-120 gpurun_out/r05/sweep_cache_2way.txt; grep -o '"eval_cache": {[^}]*}' gpurun_out/r05/sweep_cache_2way.txt
