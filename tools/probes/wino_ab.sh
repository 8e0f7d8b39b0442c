# usage: bash tools/probes/wino_ab.sh <rounds> <lib ...>   ("product" = in-tree); every lib runs with OTH_WINO=1, plus the direct kernel
rounds=$1; shift
for r in $(seq 1 $rounds); do
  unset OTHELLO_MI355X_LIB; OTH_WINO=0 python tools/netbench.py --nets 10x128x8:f16x3 2>&1 | grep -v amdgpu | sed "s|^|[direct r$r] |"
  for lib in "$@"; do
    if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
    OTH_WINO=1 python tools/netbench.py --nets 10x128x8:f16x3 2>&1 | grep -v amdgpu | sed "s|^|[wino $lib r$r] |"
  done
done
