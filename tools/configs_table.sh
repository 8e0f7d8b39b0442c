#!/bin/bash
# The reference's configurations (and BASELINE.json configs[3], configs[4]) on the streaming engine, one MI355X.
# usage (GPU box, repo root): bash tools/configs_table.sh > gpurun_out/configs_table.jsonl
set -e
B="python3 bench.py --gpus 1 --no-cpu-baseline"
$B --sims 15 --steps 4 --warmup 2 --step-games 4096                       # fast_8x8: 15 sims, 10x128
$B --sims 15 --blocks 6 --steps 4 --warmup 2 --step-games 4096            # ultrafast_8x8: 15 sims, 6x128
$B --sims 25 --steps 4 --warmup 2 --step-games 2048                       # default_8x8: 25 sims, 10x128
$B --sims 100 --steps 3 --warmup 2 --step-games 512                       # strong_8x8: 100 sims, 10x128
$B --sims 400 --steps 2 --warmup 1 --step-games 256                       # BASELINE configs[3]: 400 sims, 10x128
$B --board 6 --blocks 5 --filters 64 --sims 25 --steps 4 --warmup 3 --step-games 8192   # BASELINE configs[4]: 6x6, 5x64, 25 sims
$B --board 6 --blocks 5 --filters 64 --sims 10 --steps 4 --warmup 3 --step-games 16384  # debug_6x6.yaml: 10 sims
$B --blocks 2 --filters 16 --sims 5 --steps 4 --warmup 3 --step-games 32768             # test.yaml-sized: 2x16, 5 sims
