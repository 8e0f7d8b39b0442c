#!/bin/bash
# Round-6 evidence run on the GPU box (repo root): everything lands in gpurun_out/r06/ as small text files; the files that back a
# DESIGN claim are copied to profiles/r06_* afterwards (profiles/README.md is the index).
# usage: bash tools/collect_r06.sh [part ...]   (default: tests nets)
#   parts: smoke tests driver stats pmc nets cache exact exact3 exact4 rccl1 overlap redraw small soak longrun w6 w6pmc
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
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
  rm -rf /tmp/prof_r06
  rocprofv3 --kernel-trace --stats -d /tmp/prof_r06 -o x -- python3 bench.py --gpus 1 --steps 3 --warmup 2 --hooks-always --no-cpu-baseline --no-other-configs > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || exit 1
  db=$(find /tmp/prof_r06 -name "*.db" | head -1)
  python3 tools/rocpd_stats.py "$db" > $O/bench_kernel_stats.csv; head -4 $O/bench_kernel_stats.csv | cut -c1-150; rm -rf /tmp/prof_r06 ;;
pmc)
  tools/pmc_netbench.sh 10x128x8:f16x3 wino > $O/trunk_pmc_wino.txt 2>&1
  tools/pmc_netbench.sh 5x64x6:f16x3 w6 > $O/w6_pmc_5x64x6.txt 2>&1
  tools/bench_pmc.sh > $O/bench_pmc.txt 2>&1; cp gpurun_out/r06_bench_traffic.json $O/ ; tail -3 $O/trunk_pmc_wino.txt | cut -c1-200 ;;
nets)
  python3 tools/netbench.py 2>&1 | grep -v amdgpu > $O/netbench.log; head -3 $O/netbench.log
  OTH_WINO=0 python3 tools/netbench.py --nets 10x128x8:f16x3 2>&1 | grep -v amdgpu | sed "s/$/   [OTH_WINO=0: direct kernel k_trunk16]/" >> $O/netbench.log
  OTH_WINO6=0 python3 tools/netbench.py --nets 5x64x6:f16x3 2>&1 | grep -v amdgpu | sed "s/$/   [OTH_WINO6=0: direct kernel k_trunk_h3]/" >> $O/netbench.log ;;
cache)    # configs[1] with the opt-in evaluation cache, the driver's steps / warm-up: a labelled secondary figure, never the headline
  python3 bench.py --gpus 1 --steps 10 --warmup 3 --games 8192 --step-games 3072 --eval-cache 24 --no-cpu-baseline --no-other-configs > $O/bench_eval_cache24.json 2> $O/bench_eval_cache24.stderr.txt || exit 1
  python3 tools/print_bench_lines.py $O/bench_eval_cache24.json ;;
exact)    # the headline's own two-lane stream and the cache leg, tuple for tuple
  python3 tools/bench_stream_exact.py > $O/bench_stream_exact.log 2>&1 || exit 1; tail -1 $O/bench_stream_exact.log
  python3 tools/bench_stream_exact.py --leg cache > $O/bench_stream_exact_cache24.log 2>&1 || exit 1; tail -1 $O/bench_stream_exact_cache24.log ;;
exact3)   # item 2: bench.py's configs[3] leg at ITS shape (4608 slots, three lanes, 400 sims, one step of 255 games)
  timeout -k 10 1100 python3 tools/bench_stream_exact.py --leg configs3 > $O/leg_exact_configs3.log 2>&1; rc=$?; tail -2 $O/leg_exact_configs3.log; [ $rc -eq 0 ] || exit $rc ;;
exact4)   # item 2: the configs[4] leg at its shape (8960 slots, four lanes, 6x6 5x64, one step of 32768 games) vs the 6x6 twin
  timeout -k 10 900 python3 tools/bench_stream_exact.py --leg configs4 > $O/leg_exact_configs4.log 2>&1; rc=$?; tail -2 $O/leg_exact_configs4.log; [ $rc -eq 0 ] || exit $rc ;;
rccl1)    # item 1c: ONE full-size pass of the exchange through RCCL on the one-rank group (97 MB per step on device tensors)
  OTHELLO_FORCE_DIST=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
      bench.py --gpus 1 --steps 2 --warmup 1 --no-other-configs --no-cpu-baseline > $O/rccl1_fullsize.json 2> $O/rccl1_fullsize.stderr.txt || exit 1
  python3 tools/print_bench_lines.py $O/rccl1_fullsize.json ;;
overlap)  # item 1b: the detector's own GPU test, verbose
  python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "lane_overlap or lane_streams" > $O/lane_overlap.log 2>&1; rc=$?; tail -8 $O/lane_overlap.log; [ $rc -eq 0 ] || exit $rc ;;
redraw)   # item 1b: does drawing the streams again repair a serialised arrangement?  (same-box A/B)
  timeout -k 10 500 python3 tools/lane_redraw_ab.py > $O/lane_redraw_ab.log 2>&1; rc=$?; cat $O/lane_redraw_ab.log | cut -c1-220; [ $rc -eq 0 ] || exit $rc ;;
w6)       # item 3: the lane-group pipeline of k_trunk_w6 against the product (builds: see tools/probes/r06_w6_lgpipe.sh) + its PMC passes
  bash tools/probes/r06_w6_lgpipe.sh > /dev/null || exit 1; tail -12 $O/w6_lgpipe.log | cut -c1-200
  OTHELLO_MI355X_LIB=build/w6_lgpipe2/libothello_mi355x.so tools/pmc_netbench.sh 5x64x6:f16x3 w6lg > $O/w6_lgpipe_pmc.txt 2>&1; tail -4 $O/w6_lgpipe_pmc.txt | cut -c1-200 ;;
w6pmc)    # only the PMC passes of the pipeline build
  OTHELLO_MI355X_LIB=build/w6_lgpipe2/libothello_mi355x.so tools/pmc_netbench.sh 5x64x6:f16x3 w6lg > $O/w6_lgpipe_pmc.txt 2>&1; tail -4 $O/w6_lgpipe_pmc.txt | cut -c1-200 ;;
soak)     # the wide parity sweep and the full-size exact run on the final build
  python3 tools/parity_sweep.py > $O/parity_sweep.log 2>&1; rc=$?; tail -1 $O/parity_sweep.log; [ $rc -eq 0 ] || exit $rc
  python3 tools/fullsize_exact.py > $O/fullsize_exact.log 2>&1; rc=$?; tail -1 $O/fullsize_exact.log; [ $rc -eq 0 ] || exit $rc ;;
longrun)  # stability: 120 timed steps of the headline (~5.5 minutes of steady state), no legs, no CPU baseline
  python3 bench.py --gpus 1 --steps 120 --warmup 5 --no-other-configs --no-cpu-baseline > $O/soak_120steps.json 2> $O/soak_120steps.stderr.txt || exit 1
  python3 tools/print_bench_lines.py $O/soak_120steps.json ;;
small)    # item 5: the drop-in at the reference's own call sizes, batch and continuous mode (INTEGRATION.md section 1's table)
  timeout -k 10 400 python3 tools/small_config_rate.py 2>&1 | grep -v amdgpu > $O/small_config.log; rc=$?; cat $O/small_config.log; [ $rc -eq 0 ] || exit $rc ;;
esac; done
