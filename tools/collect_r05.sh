#!/bin/bash
# Round-5 evidence run on the GPU box (repo root): everything lands in gpurun_out/r05/ as small text files.
# usage: bash tools/collect_r05.sh [part ...]   parts: tests driver stats pmc nets cache exact probe   (default: tests nets)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
parts=${@:-tests nets}
for part in $parts; do case $part in
smoke)    # the driver's smoke entry
  python3 __graft_entry__.py --smoke > $O/smoke.log 2>&1; rc=$?; tail -2 $O/smoke.log; [ $rc -eq 0 ] || exit $rc ;;
tests)    # the whole GPU suite, incl. the three child stages (self-launched gloo rehearsal, RCCL one-rank bench and worker)
  python3 -m pytest tests -m gpu -x -q -s > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc $rc" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
  [ $rc -eq 0 ] || exit $rc ;;
driver)   # the driver's own command, verbatim (headline + the other_configs legs + CPU baseline)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.stderr.txt || exit 1
  python3 tools/print_bench_lines.py $O/bench_driver_cmd.json ;;
stats)    # rocprofv3 --kernel-trace --stats of bench.py with the hooks on everywhere (same launches timed by both)
  rm -rf /tmp/prof_r05
  rocprofv3 --kernel-trace --stats -d /tmp/prof_r05 -o x -- python3 bench.py --gpus 1 --steps 3 --warmup 2 --hooks-always --no-cpu-baseline --no-other-configs > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || exit 1
  db=$(find /tmp/prof_r05 -name "*.db" | head -1)
  python3 tools/rocpd_stats.py "$db" > $O/bench_kernel_stats.csv; head -4 $O/bench_kernel_stats.csv | cut -c1-150; rm -rf /tmp/prof_r05 ;;
pmc)
  tools/pmc_netbench.sh 10x128x8:f16x3 wino > $O/trunk_pmc_wino.txt 2>&1
  tools/pmc_netbench.sh 5x64x6:f16x3 w6 > $O/w6_pmc_5x64x6.txt 2>&1
  tools/bench_pmc.sh > $O/bench_pmc.txt 2>&1; cp gpurun_out/r05_bench_traffic.json $O/ ; tail -3 $O/trunk_pmc_wino.txt | cut -c1-200 ;;
nets)
  python3 tools/netbench.py 2>&1 | grep -v amdgpu > $O/netbench.log; head -3 $O/netbench.log
  OTH_WINO=0 python3 tools/netbench.py --nets 10x128x8:f16x3 2>&1 | grep -v amdgpu | sed "s/$/   [OTH_WINO=0: direct kernel k_trunk16]/" >> $O/netbench.log
  OTH_WINO6=0 python3 tools/netbench.py --nets 5x64x6:f16x3 2>&1 | grep -v amdgpu | sed "s/$/   [OTH_WINO6=0: direct kernel k_trunk_h3]/" >> $O/netbench.log ;;
cache)    # configs[1] with the opt-in evaluation cache, the driver's steps / warm-up: a labelled secondary figure, never the headline
  python3 bench.py --gpus 1 --steps 10 --warmup 3 --games 8192 --step-games 3072 --eval-cache 24 --no-cpu-baseline --no-other-configs > $O/bench_eval_cache24.json 2> $O/bench_eval_cache24.stderr.txt || exit 1
  python3 tools/print_bench_lines.py $O/bench_eval_cache24.json ;;
exact)
  python3 tools/bench_stream_exact.py > $O/bench_stream_exact.log 2>&1 || exit 1; tail -1 $O/bench_stream_exact.log
  OTH_EXACT_CACHE=24 OTH_EXACT_SLOTS=8192 python3 tools/bench_stream_exact.py 2 3072 > $O/bench_stream_exact_cache24.log 2>&1 || exit 1; tail -1 $O/bench_stream_exact_cache24.log ;;
esac; done
