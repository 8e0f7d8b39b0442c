#!/bin/bash
# Interleaved A/B of trunk builds in one GPU session: tools/ab_netbench.sh <rounds> <nets> <lib1> <lib2> ...
# ("product" = the in-tree library).  One netbench process per (round, lib); compare medians, same box only.
rounds=$1; nets=$2; shift 2
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
    python tools/netbench.py --nets "$nets" 2>&1 | grep -v amdgpu.ids | sed "s|^|[$lib r$r] |"
  done
done
