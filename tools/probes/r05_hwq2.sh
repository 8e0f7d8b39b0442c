set -e
mkdir -p gpurun_out/r05; L=gpurun_out/r05/hw_queues_b.log; : > $L
export GPU_MAX_HW_QUEUES=8
echo "== GPU_MAX_HW_QUEUES=8: configs[4] lanes" >> $L
timeout -k 10 200 python3 tools/leg_sweep.py configs4 8960:4 11200:5 13440:6 8960:4 2>/dev/null | cut -c1-200 >> $L
echo "== GPU_MAX_HW_QUEUES=8: headline lanes" >> $L
timeout -k 10 300 python3 tools/leg_sweep.py headline 4096:2 4096:4 4098:3::1536 4096:2 2>/dev/null | cut -c1-200 >> $L
cat $L
