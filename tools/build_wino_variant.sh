#!/bin/bash
# A/B variant of net_wino.hip only (the other objects come from the product build): tools/build_wino_variant.sh <name> [hipcc flags]
# -> build/<name>/libothello_mi355x.so, selected with OTHELLO_MI355X_LIB
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd); csrc=$root/othello_reinforcement_learning_test_amd/csrc; out=$root/build/$name; mkdir -p $out
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $csrc/net_wino.hip -o $out/net_wino.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libothello_mi355x.so $out/net_wino.o $csrc/net_mfma.o $csrc/net_h3.o $csrc/net_wino6.o $csrc/rules_api.o $csrc/net.o $csrc/net_f32.o $csrc/engine.o $csrc/replay_ops.o
rm -f $out/net_wino.o; echo built $out
