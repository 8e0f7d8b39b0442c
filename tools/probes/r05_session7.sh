set -e
mkdir -p gpurun_out/r05
timeout -k 10 300 python3 tools/leg_sweep.py configs3 4608:3::255 4416:2 > gpurun_out/r05/sweep_configs3_b.txt 2> gpurun_out/r05/sweep_configs3_b.err
cat gpurun_out/r05/sweep_configs3_b.txt
timeout -k 10 300 python3 tools/leg_sweep.py cache 8192:2::3072 8192:2::3072::23 8192:2::3072::24 8192:2::4096::24 > gpurun_out/r05/sweep_cache_b.txt 2> gpurun_out/r05/sweep_cache_b.err
cat gpurun_out/r05/sweep_cache_b.txt
timeout -k 10 120 python3 tools/leg_sweep.py configs4 6144:3 6144:3 > gpurun_out/r05/sweep_configs4_base.txt 2> gpurun_out/r05/sweep_configs4_base.err
cat gpurun_out/r05/sweep_configs4_base.txt
