# usage: bash tools/probes/nets_ab.sh <rounds> <nets spec> <lib ...>   ("product" = in-tree): netbench A/B on one box, interleaved
rounds=$1; nets=$2; shift; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
    python tools/netbench.py --nets $nets 2>&1 | grep -v amdgpu | sed "s|^|[$lib r$r] |"
  done
done
