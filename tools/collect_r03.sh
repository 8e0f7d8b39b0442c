#!/bin/bash
# Round-3 evidence run on the GPU box (repo root): everything lands in gpurun_out/r03/ as small text files.
# usage: bash tools/collect_r03.sh [part ...]   parts: driver stats pmc nets configs small exact   (default: all)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
parts=${@:-driver stats pmc nets configs small exact}
for part in $parts; do case $part in
driver)   # the driver's own command, verbatim
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.stderr.txt
  tail -c 600 $O/bench_driver_cmd.json | head -c 300; echo ;;
stats)    # rocprofv3 --kernel-trace --stats of bench.py with the hooks on everywhere (same launches timed by both)
  rm -rf /tmp/prof_r03
  rocprofv3 --kernel-trace --stats -d /tmp/prof_r03 -o x -- python3 bench.py --gpus 1 --steps 3 --warmup 2 --hooks-always --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
  db=$(find /tmp/prof_r03 -name "*.db" | head -1)
  python3 tools/rocpd_stats.py "$db" > $O/bench_kernel_stats.csv; head -4 $O/bench_kernel_stats.csv | cut -c1-150; rm -rf /tmp/prof_r03 ;;
pmc)
  tools/pmc_netbench.sh 10x128x8:f16x3 wino > $O/trunk_pmc_wino.txt 2>&1
  OTH_WINO=0 tools/pmc_netbench.sh 10x128x8:f16x3 k16_tp2 > $O/trunk_pmc_tp2.txt 2>&1
  OTH_WINO=0 OTH_TRUNK_TP=4 tools/pmc_netbench.sh 10x128x8:f16x3 k16_tp4 > $O/trunk_pmc_tp4.txt 2>&1
  tools/pmc_netbench.sh 5x64x6:f16x3 h3_6 > $O/h3_pmc_5x64x6.txt 2>&1
  tools/pmc_netbench.sh 5x64x8:f16x3 h3_8 > $O/h3_pmc_5x64x8.txt 2>&1
  tools/bench_pmc.sh > $O/bench_pmc.txt 2>&1; cp gpurun_out/r03_bench_traffic.json $O/ ; tail -3 $O/trunk_pmc_wino.txt | cut -c1-200 ;;
nets)
  python3 tools/netbench.py 2>&1 | grep -v amdgpu > $O/netbench.log; head -3 $O/netbench.log
  OTH_WINO=0 python3 tools/netbench.py --nets 10x128x8:f16x3 2>&1 | grep -v amdgpu | sed "s/$/   [OTH_WINO=0: direct kernel k_trunk16]/" >> $O/netbench.log ;;
configs)
  bash tools/configs_table.sh > $O/configs_table.jsonl 2> $O/configs_table.err; python3 tools/print_bench_lines.py $O/configs_table.jsonl ;;
small)
  python3 tools/smallg.py > $O/smallg.log 2>&1; python3 tools/small_config_rate.py > $O/small_config.log 2>&1; python3 tools/dropin_rate.py > $O/dropin.log 2>&1; tail -2 $O/smallg.log ;;
exact)
  python3 tools/bench_stream_exact.py > $O/bench_stream_exact.log 2>&1; tail -2 $O/bench_stream_exact.log
  python3 tools/parity_sweep.py 60 > $O/parity_sweep.log 2>&1; tail -2 $O/parity_sweep.log ;;
esac; done
