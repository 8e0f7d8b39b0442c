# usage: bash tools/probes/wino_ab3.sh <rounds> <lib ...>   ("product" = in-tree): Winograd trunk only, interleaved rounds
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
    python tools/netbench.py --nets 10x128x8:f16x3 2>&1 | grep -v amdgpu | sed "s|^|[$lib r$r] |"
  done
done
