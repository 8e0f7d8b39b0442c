#!/bin/bash
# Round 6, VERDICT r5 item 3: the lane-group pipeline of k_trunk_w6 (tools/probes/w6_lgpipe_loop.inc), measured against the
# product kernel on ONE box: bit-identity of the outputs, interleaved netbench, in-kernel stamps, the configs[4] leg.
# builds (in the container; U="-mllvm -pragma-unroll-threshold=4000000"):
#   tools/build_variant.sh w6_stamps         --only net_wino6 -DOTH_STAMPS
#   tools/build_variant.sh w6_lgpipe2        --ablation w6_exp_lgpipe --only net_wino6 $U            (micro-op form, two-group ring)
#   tools/build_variant.sh w6_lgpipe2_stamps --ablation w6_exp_lgpipe --only net_wino6 $U -DOTH_STAMPS
#   tools/build_variant.sh w6_lgpipe3[_stamps] ... $U -DOTH_W6_RING=3 [-DOTH_STAMPS]                  (three-group ring)
# (the first form -- sched_group_barrier pattern, profiles/r06_w6_lgpipe_sgb.log -- is commit 6ad2894's w6_lgpipe_loop.inc)
# usage (GPU box): W6_LIBS="w6_lgpipe2 w6_lgpipe3" W6_STAMP_LIBS="w6_lgpipe2_stamps w6_lgpipe3_stamps" bash tools/probes/r06_w6_lgpipe.sh
set -e
O=gpurun_out/r06; mkdir -p $O
L=$O/w6_lgpipe.log; : > $L
libs="product ${W6_LIBS:-w6_lgpipe2}"
echo "== outputs on 4099 fixed positions (5x64 on 6x6): every variant against the product build, bit for bit" >> $L
for lib in $libs; do
  if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
  timeout -k 10 120 python3 tools/probes/w6_dump_outputs.py /tmp/w6_$lib.npz 2>&1 | grep -v amdgpu >> $L
done
unset OTHELLO_MI355X_LIB
for lib in $libs; do [ "$lib" = product ] || python3 tools/probes/w6_dump_outputs.py /tmp/w6_product.npz /tmp/w6_$lib.npz >> $L 2>&1 || true; done
echo "== interleaved netbench, 5x64 on 6x6, 4096 positions per launch (ms per launch; max error vs torch fp32)" >> $L
bash tools/ab_netbench.sh 3 5x64x6:f16x3 $libs >> $L 2>&1
echo "== in-kernel stamps (per-wave cycles; diagnostic builds)" >> $L
for lib in w6_stamps ${W6_STAMP_LIBS:-w6_lgpipe2_stamps}; do
  OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so python3 tools/netbench.py --nets 5x64x6:f16x3 2>&1 | grep -E "stamps|ms /" | tail -2 | sed "s|^|[$lib] |" >> $L
done
echo "== configs[4] leg (8960 games in four lanes), games/s" >> $L
for r in 1 2; do for lib in $libs; do
  if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
  timeout -k 10 120 python3 tools/leg_sweep.py configs4 8960:4 2>/dev/null | sed "s|^|[$lib r$r] |" >> $L
done; done
unset OTHELLO_MI355X_LIB
cut -c1-260 $L
