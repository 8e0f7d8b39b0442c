#!/bin/bash
# Build an A/B variant of libothello_mi355x.so: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DOTH_STAMPS]
# Only net_mfma.hip / net_h3.hip are recompiled with the extra flags; the other objects come from the product build.
# Output: build/<name>/libothello_mi355x.so (travels to the GPU box; select it with OTHELLO_MI355X_LIB=...).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/othello_reinforcement_learning_test_amd/csrc
out=$root/build/$name
mkdir -p "$out"
make -s -j4 -C "$csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -Wno-unused-result"
for f in net_mfma net_h3 net_wino net_wino6; do  # (heads experiments: net_h3 includes net_heads_wave.h)
  /opt/rocm/bin/hipcc $FLAGS "$@" -c "$csrc/$f.hip" -o "$out/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libothello_mi355x.so" "$out/net_mfma.o" "$out/net_h3.o" "$out/net_wino.o" "$out/net_wino6.o" \
  "$csrc/rules_api.o" "$csrc/net.o" "$csrc/net_f32.o" "$csrc/engine.o" "$csrc/replay_ops.o"
rm -f "$out"/*.o
echo "built $out/libothello_mi355x.so"
