#!/bin/bash
# Does the lane-overlap check of the first warm-up step (HIP-event hooks on for that step only) cost the timed region anything?
# The driver's headline command without legs / CPU baseline, with and without the check, interleaved on ONE box.
O=gpurun_out/r06; mkdir -p $O; L=$O/lane_check_ab.log; : > $L
for r in 1 2; do for mode in check nocheck; do
  if [ $mode = nocheck ]; then export OTHELLO_NO_LANE_CHECK=1; else unset OTHELLO_NO_LANE_CHECK; fi
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('[$mode r$r] %.1f games/s | launch %.4f ms | frac %.4f | overlap %s | warm-up check %s' % (d['value'], r['avg_launch_ms'], r['frac'], r['lanes_overlap'], r['lanes_check_warmup'].get('lanes_overlap')))" >> $L
done; done
unset OTHELLO_NO_LANE_CHECK
cat $L
