# usage: bash tools/probes/bench_ab.sh <rounds> <lib ...>   ("product" = in-tree): bench.py A/B on one box, interleaved
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ "$lib" = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
    python bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | python tools/print_bench_lines.py /dev/stdin | sed "s|^|[$lib r$r] |"
  done
done
