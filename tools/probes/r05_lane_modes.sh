set -e
mkdir -p gpurun_out/r05; L=gpurun_out/r05/lane_modes.log; : > $L
echo "== three identical two-lane cache legs in ONE process, new streams per workload (behaviour up to round 5)" >> $L
OTHELLO_BENCH_NEW_STREAMS=1 timeout -k 10 300 python3 tools/leg_sweep.py cache 8192:2::3072::24 8192:2::3072::24 8192:2::3072::24 2>/dev/null | cut -c1-250 >> $L
echo "== the same with the lanes' streams made once per process" >> $L
timeout -k 10 300 python3 tools/leg_sweep.py cache 8192:2::3072::24 8192:2::3072::24 8192:2::3072::24 2>/dev/null | cut -c1-250 >> $L
cat $L
