#!/bin/bash
# Builds of the value-head FC1 form that gave wrong values in k_trunk_h3<32, 8, 1, 4> (net_heads_wave.h, OTH_HEADS_BATCH4):
#   1 = as it was (commit 42d30f7^: the kernel's ISA is identical to that commit's, label names aside)
#   2 = the FMAs pinned scalar (no v_pk_fma_f32)      3 = 4 rows per batch (16 loads in flight instead of 32)
#   4 = biases added after the loop (accumulators are not destinations of loads that are still in flight)
#   5 = s_waitcnt vmcnt(0) between the batch's loads and its FMAs
# usage (CPU box): bash tools/probes/heads_batch4_variants.sh    -> build/hb4_<v>/libothello_mi355x.so
# then on the GPU box: for v in 1 2 3 4 5; do OTHELLO_MI355X_LIB=build/hb4_$v/libothello_mi355x.so python tools/probes/diag_v3.py; done
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
csrc=$root/othello_reinforcement_learning_test_amd/csrc
make -s -j4 -C "$csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -Wno-unused-result"
for v in ${VARIANTS:-1 2 3 4 5}; do
  out=$root/build/hb4_$v; mkdir -p "$out"
  ( /opt/rocm/bin/hipcc $FLAGS -DOTH_HEADS_BATCH4=$v -c "$csrc/net_h3.hip" -o "$out/net_h3.o" &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libothello_mi355x.so" "$out/net_h3.o" "$csrc/net_mfma.o" \
      "$csrc/net_wino.o" "$csrc/net_wino6.o" "$csrc/rules_api.o" "$csrc/net.o" "$csrc/net_f32.o" "$csrc/engine.o" "$csrc/replay_ops.o" &&
    rm -f "$out/net_h3.o" && echo "built $out/libothello_mi355x.so" ) &
done
wait
