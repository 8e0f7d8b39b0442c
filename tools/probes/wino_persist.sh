#!/bin/bash
# Round-4 experiment: persistent workgroups for k_trunk_w -- OTH_WINO_PERSIST=g launches g looping
# workgroups instead of one per position pair.  Timing (interleaved, 4096 positions per launch) and the memory-side traffic
# (FETCH_SIZE, separate --pmc pass as MI355X_MICROARCH.md prescribes) of the default launch and of g = 256 / 512.
# usage (GPU box, repo root): bash tools/probes/wino_persist.sh > gpurun_out/wino_persist.txt
export TMPDIR=/tmp
for r in 1 2 3; do
  for g in 0 256 512; do
    OTH_WINO_PERSIST=$g python3 tools/netbench.py --nets 10x128x8:f16x3 2>&1 | grep -v amdgpu.ids | sed "s|^|[persist $g r$r] |"
  done
done
for g in 0 256; do
  out=gpurun_out/pmc_persist_$g; rm -rf $out; mkdir -p $out
  OTH_WINO_PERSIST=$g rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python3 tools/netbench.py --nets 10x128x8:f16x3 > $out/fetch.log 2>&1
  echo "[persist $g] $(python3 tools/pmc_summary.py $out/fetch | sed "s|^$out/||")"
  OTH_WINO_PERSIST=$g rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/tcc -o tcc -- python3 tools/netbench.py --nets 10x128x8:f16x3 > $out/tcc.log 2>&1
  echo "[persist $g] $(python3 tools/pmc_summary.py $out/tcc | sed "s|^$out/||")"
  rm -rf $out
done
