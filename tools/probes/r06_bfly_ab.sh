#!/bin/bash
# Round 6: the register-to-register wave reductions (csrc/wave_bfly.h) against the build before them (build/pre_bfly = the
# library with the __shfl_xor butterflies): outputs of four networks bit for bit, interleaved netbench, small engines (tree-bound),
# the configs[4] leg.   usage (GPU box): bash tools/probes/r06_bfly_ab.sh   -> profiles/r06_bfly_ab.log
O=gpurun_out/r06; mkdir -p $O
L=$O/bfly_ab.log; : > $L
timeout -k 10 60 build/probe_bfly >> $L 2>&1
echo "== network outputs, 4099 fixed positions: this build vs build/pre_bfly" >> $L
for net in 5x64x6: 10x128x8: 5x64x8: 2x16x8:f32 2x32x6:; do
  n=${net%%:*}; pr=${net##*:}
  for lib in product pre_bfly; do
    if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
    DUMP_NET=$n DUMP_PREC=$pr timeout -k 10 120 python3 tools/probes/w6_dump_outputs.py /tmp/bf_${n}_$lib.npz 2>&1 | grep -v amdgpu | sed "s|^|[$n] |" >> $L
  done
  unset OTHELLO_MI355X_LIB
  python3 tools/probes/w6_dump_outputs.py /tmp/bf_${n}_pre_bfly.npz /tmp/bf_${n}_product.npz | sed "s|^|[$n] |" >> $L 2>&1
done
echo "== interleaved netbench (ms per 4096 positions)" >> $L
bash tools/ab_netbench.sh 3 5x64x6:f16x3,10x128x8:f16x3,2x32x8:f16x3 pre_bfly product >> $L 2>&1
echo "== small engines (tools/smallg.py: tree kernel in the open), pre_bfly then product, twice" >> $L
for r in 1 2; do for lib in pre_bfly product; do
  if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
  timeout -k 10 120 python3 tools/smallg.py 2>&1 | grep "G=" | sed "s|^|[$lib r$r] |" >> $L
done; done
echo "== configs[4] leg (8960 games in four lanes), games/s" >> $L
for r in 1 2; do for lib in pre_bfly product; do
  if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
  timeout -k 10 120 python3 tools/leg_sweep.py configs4 8960:4 2>/dev/null | cut -c1-230 | sed "s|^|[$lib r$r] |" >> $L
done; done
unset OTHELLO_MI355X_LIB
cut -c1-200 $L
