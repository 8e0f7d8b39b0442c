set -e
mkdir -p gpurun_out/r05; L=gpurun_out/r05/hw_queues.log; : > $L
for q in default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "== GPU_MAX_HW_QUEUES=$q: configs[4] 8960:4 twice, configs[3] 4608:3, headline 4096:2" >> $L
  timeout -k 10 120 python3 tools/leg_sweep.py configs4 8960:4 8960:4 2>/dev/null | cut -c1-200 >> $L
  timeout -k 10 200 python3 tools/leg_sweep.py configs3 4608:3::255 2>/dev/null | cut -c1-200 >> $L
  timeout -k 10 120 python3 tools/leg_sweep.py headline 4096:2 2>/dev/null | cut -c1-200 >> $L
done
cat $L
