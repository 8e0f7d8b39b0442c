# usage: bash tools/probes/ring_ab.sh <lib ...>: one-position / small-batch trunk launches (k_trunk_w<1>) per build
for lib in "$@"; do
  if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
  for n in 1 32 256; do python tools/netbench.py --n $n --nets 10x128x8:f16x3 2>&1 | grep -v amdgpu | sed "s|^|[$lib n=$n] |"; done
done
