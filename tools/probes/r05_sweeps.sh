#!/bin/bash
# The GPU sessions behind round 5's leg sizes and probes (each block was one gpurun call; outputs copied to profiles/r05_*):
#   probe    the extended operand-select probe (ONE run)                      -> profiles/r05_probe_pk_opsel.txt
#   configs3 slots / lanes of the configs[3] leg                              -> profiles/r05_sweep_configs3.txt, _b.txt
#   cache    slots / step / table size of the evaluation-cache leg            -> profiles/r05_sweep_cache.txt, _b.txt
#   configs4 slots / lanes of the configs[4] leg                              -> profiles/r05_sweep_configs4_base.txt, _b.txt
#   soak     parity sweep + full-size exact run on the final build            -> profiles/r05_parity_sweep.log, r05_fullsize_exact.log
#   lanes    three identical two-lane legs in one process, new streams per workload vs the per-process pool
#                                                                             -> profiles/r05_lane_modes.log
#   hwq      GPU_MAX_HW_QUEUES default (4) vs 8 on the legs and the headline; lane counts under 8 queues
#                                                                             -> profiles/r05_hw_queues.log, r05_hw_queues_b.log
#   cache2way  the two-way cache: its tests and a table-size sweep             -> profiles/r05_sweep_cache_2way.txt
# usage (GPU box, repo root): bash tools/probes/r05_sweeps.sh probe|configs3|cache|configs4|soak|lanes|hwq|cache2way
set -e
O=gpurun_out/r05; mkdir -p $O
case "$1" in
probe)
  hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o build/probe_pk_opsel2 tools/probes/probe_pk_opsel2.hip
  timeout -k 10 180 build/probe_pk_opsel2 > $O/probe_pk_opsel.txt 2>&1; cat $O/probe_pk_opsel.txt ;;
configs3)
  timeout -k 10 420 python3 tools/leg_sweep.py configs3 2048:2 4096:2 4416:2 4608:3 6624:3 > $O/sweep_configs3.txt
  timeout -k 10 300 python3 tools/leg_sweep.py configs3 4608:3::255 4416:2 > $O/sweep_configs3_b.txt; cat $O/sweep_configs3*.txt ;;
cache)
  timeout -k 10 300 python3 tools/leg_sweep.py cache 4096:2 8192:2 8192:2::3072 12288:3::3072 > $O/sweep_cache.txt
  timeout -k 10 300 python3 tools/leg_sweep.py cache 8192:2::3072 8192:2::3072::23 8192:2::3072::24 8192:2::4096::24 > $O/sweep_cache_b.txt; cat $O/sweep_cache*.txt ;;
configs4)
  timeout -k 10 120 python3 tools/leg_sweep.py configs4 6144:3 6144:3 > $O/sweep_configs4_base.txt
  timeout -k 10 400 python3 tools/leg_sweep.py configs4 6144:3 8192:4 6720:3 6600:3 8960:4 10240:5 > $O/sweep_configs4_b.txt; cat $O/sweep_configs4*.txt ;;
soak)
  python3 tools/parity_sweep.py > $O/parity_sweep.log 2>&1; tail -1 $O/parity_sweep.log
  python3 tools/fullsize_exact.py > $O/fullsize_exact.log 2>&1; tail -1 $O/fullsize_exact.log ;;
lanes)
  L=$O/lane_modes.log; : > $L
  echo "== three identical two-lane cache legs in ONE process, new streams per workload (behaviour up to round 5)" >> $L
  OTHELLO_BENCH_NEW_STREAMS=1 timeout -k 10 300 python3 tools/leg_sweep.py cache 8192:2::3072::24 8192:2::3072::24 8192:2::3072::24 2>/dev/null | cut -c1-250 >> $L
  echo "== the same with the lanes' streams made once per process" >> $L
  timeout -k 10 300 python3 tools/leg_sweep.py cache 8192:2::3072::24 8192:2::3072::24 8192:2::3072::24 2>/dev/null | cut -c1-250 >> $L
  cat $L ;;
hwq)
  L=$O/hw_queues.log; : > $L
  for q in default 8; do
    if [ $q = default ]; then export GPU_MAX_HW_QUEUES=4; else export GPU_MAX_HW_QUEUES=$q; fi   # (bench.py / the package default to 8 since round 5)
    echo "== GPU_MAX_HW_QUEUES=$q: configs[4] 8960:4 twice, configs[3] 4608:3, headline 4096:2" >> $L
    timeout -k 10 120 python3 tools/leg_sweep.py configs4 8960:4 8960:4 2>/dev/null | cut -c1-200 >> $L
    timeout -k 10 200 python3 tools/leg_sweep.py configs3 4608:3::255 2>/dev/null | cut -c1-200 >> $L
    timeout -k 10 120 python3 tools/leg_sweep.py headline 4096:2 2>/dev/null | cut -c1-200 >> $L
  done
  export GPU_MAX_HW_QUEUES=8; L2=$O/hw_queues_b.log; : > $L2
  echo "== GPU_MAX_HW_QUEUES=8: configs[4] lanes" >> $L2
  timeout -k 10 200 python3 tools/leg_sweep.py configs4 8960:4 11200:5 13440:6 8960:4 2>/dev/null | cut -c1-200 >> $L2
  echo "== GPU_MAX_HW_QUEUES=8: headline lanes" >> $L2
  timeout -k 10 300 python3 tools/leg_sweep.py headline 4096:2 4096:4 4098:3::1536 4096:2 2>/dev/null | cut -c1-200 >> $L2
  cat $L $L2 ;;
cache2way)
  python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_selfplay_exact.py tests/test_gpu_multirank.py -m gpu -x -q -k "cache" 2>&1 | tail -3
  timeout -k 10 500 python3 tools/leg_sweep.py cache 8192:2::3072::22 8192:2::3072::23 8192:2::3072::24 8192:2::3072::22 8192:2::3072::24 > $O/sweep_cache_2way.txt
  cut -c1-120 $O/sweep_cache_2way.txt ;;
*) echo "usage: $0 probe|configs3|cache|configs4|soak|lanes|hwq|cache2way"; exit 2 ;;
esac
