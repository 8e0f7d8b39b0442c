#!/bin/bash
# Same-box A/B of this round's k_trunk_w (run-time activation scale) against round 4's: build/r4wino/libothello_mi355x.so = the
# net_wino.hip of commit fa3db74 (plus the one-line helper wino_positions_per_workgroup) compiled with the product flags and linked
# against the current objects.  -> profiles/r05_ab_r4wino.log (2.534 vs 2.538 ms per launch, 601 vs 602 games/s)
set -e
O=gpurun_out/r05; mkdir -p $O; L=$O/ab_r4wino.log; : > $L
echo "== interleaved netbench: this round's k_trunk_w (run-time activation scale) vs round 4's (build/r4wino: fa3db74's net_wino.hip against the current library)" >> $L
bash tools/ab_netbench.sh 4 10x128x8:f16x3 product r4wino >> $L 2>&1
echo "== interleaved bench.py (6 timed steps)" >> $L
for r in 1 2; do for lib in product r4wino; do
  if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
  python3 bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$lib r$r] %.1f games/s, frac %.4f, launch %.4f ms' % (d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms']))" >> $L
done; done
cat $L
