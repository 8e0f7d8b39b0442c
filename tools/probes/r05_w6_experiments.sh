#!/bin/bash
# Round 5, VERDICT r4 item 4: the two experiments on k_trunk_w6's idle-matrix-pipe share, interleaved A/B on ONE box.
#   product          the shipped kernel
#   w6_fc_lds        (a') the heads' FC weights staged through LDS once per workgroup (bit-identical outputs)
#   w6_early         (b)  the finished accumulators read into VGPR copies between the last group's MFMAs
# builds: tools/build_variant.sh <name> --ablation w6_exp_fc_lds|w6_exp_early_acc_reads --only net_wino6 [-DOTH_STAMPS]
set -e
O=gpurun_out/r05; mkdir -p $O
L=$O/w6_experiments.log; : > $L
echo "== interleaved netbench, 5x64 on 6x6, 4096 positions per launch (ms per launch; max error vs torch fp32)" >> $L
bash tools/ab_netbench.sh 3 5x64x6:f16x3 product w6_fc_lds w6_early >> $L 2>&1
echo "== in-kernel stamps (per-wave cycles; diagnostic builds)" >> $L
for lib in w6_stamps w6_fc_lds_stamps w6_early_stamps; do
  OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so python3 tools/netbench.py --nets 5x64x6:f16x3 2>&1 | grep -E "stamps|ms /" | tail -2 | sed "s|^|[$lib] |" >> $L
done
echo "== configs[4] leg (8960 games in four lanes), games/s" >> $L
for r in 1 2; do for lib in product w6_fc_lds w6_early; do
  if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
  python3 tools/leg_sweep.py configs4 8960:4 2>/dev/null | sed "s|^|[$lib r$r] |" >> $L
done; done
unset OTHELLO_MI355X_LIB
cat $L
