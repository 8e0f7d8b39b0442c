#!/bin/bash
# The GPU sessions behind round 5's leg sizes and probes (each block was one gpurun call; outputs copied to profiles/r05_*):
#   probe    the extended operand-select probe (ONE run)                      -> profiles/r05_probe_pk_opsel.txt
#   configs3 slots / lanes of the configs[3] leg                              -> profiles/r05_sweep_configs3.txt, _b.txt
#   cache    slots / step / table size of the evaluation-cache leg            -> profiles/r05_sweep_cache.txt, _b.txt
#   configs4 slots / lanes of the configs[4] leg                              -> profiles/r05_sweep_configs4_base.txt, _b.txt
#   soak     parity sweep + full-size exact run on the final build            -> profiles/r05_parity_sweep.log, r05_fullsize_exact.log
# usage (GPU box, repo root): bash tools/probes/r05_sweeps.sh probe|configs3|cache|configs4|soak
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
*) echo "usage: $0 probe|configs3|cache|configs4|soak"; exit 2 ;;
esac
