#!/bin/bash
# Build an A/B variant of libothello_mi355x.so:
#   tools/build_variant.sh <name> [--ablation NAME]... [--only "net_wino6 ..."] [extra hipcc flags, e.g. -DOTH_STAMPS]
# The trunk sources (net_mfma / net_h3 / net_wino / net_wino6, or the --only list) are compiled from a COPY of csrc/ with the
# extra flags -- after the edits of tools/probes/apply_ablation.py for every --ablation (timing ablations: WRONG results by
# construction, which is why they are not `-D` switches of the product sources) -- and linked against the product's other objects.
# Output: build/<name>/libothello_mi355x.so (travels to the GPU box; select it with OTHELLO_MI355X_LIB=...).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/othello_reinforcement_learning_test_amd/csrc
out=$root/build/$name
files="net_mfma net_h3 net_wino net_wino6"
abl=()
while [ $# -gt 0 ]; do
  case "$1" in
    --ablation) abl+=("$2"); shift 2;;
    --only) files="$2"; shift 2;;
    *) break;;
  esac
done
mkdir -p "$out"
make -s -j4 -C "$csrc"
tmp=$(mktemp -d); src=$tmp/pkg/csrc   # common.h includes "../../include/othello_mi355x.h"
mkdir -p "$src" "$tmp/include"
cp "$csrc"/*.hip "$csrc"/*.h "$src"/
cp "$root/include/othello_mi355x.h" "$tmp/include/"
for a in "${abl[@]}"; do python3 "$root/tools/probes/apply_ablation.py" "$a" "$src"; done
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -Wno-unused-result"
objs=""
for f in net_mfma net_h3 net_wino net_wino6; do
  if [[ " $files " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc $FLAGS "$@" -c "$src/$f.hip" -o "$out/$f.o" &
    objs="$objs $out/$f.o"
  else
    objs="$objs $csrc/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libothello_mi355x.so" $objs \
  "$csrc/rules_api.o" "$csrc/net.o" "$csrc/net_f32.o" "$csrc/engine.o" "$csrc/replay_ops.o"
rm -f "$out"/*.o; rm -rf "$tmp"
echo "built $out/libothello_mi355x.so"
