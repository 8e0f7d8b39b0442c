set -e
mkdir -p gpurun_out/r05
timeout -k 10 400 python3 tools/leg_sweep.py configs4 6144:3 8192:4 6720:3 6600:3 8960:4 10240:5 > gpurun_out/r05/sweep_configs4_b.txt 2> gpurun_out/r05/sweep_configs4_b.err
cat gpurun_out/r05/sweep_configs4_b.txt
