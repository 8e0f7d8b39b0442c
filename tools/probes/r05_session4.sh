set -e
mkdir -p gpurun_out/r05
timeout -k 10 180 build/probe_pk_opsel2 > gpurun_out/r05/probe_pk_opsel.txt 2>&1
tail -13 gpurun_out/r05/probe_pk_opsel.txt
timeout -k 10 420 python3 tools/leg_sweep.py configs3 2048:2 4096:2 4416:2 4608:3 6624:3 > gpurun_out/r05/sweep_configs3.txt 2> gpurun_out/r05/sweep_configs3.err
cat gpurun_out/r05/sweep_configs3.txt
timeout -k 10 300 python3 tools/leg_sweep.py cache 4096:2 8192:2 8192:2::3072 12288:3::3072 > gpurun_out/r05/sweep_cache.txt 2> gpurun_out/r05/sweep_cache.err
cat gpurun_out/r05/sweep_cache.txt
