#!/bin/bash
# rocprofv3 --pmc passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes) over tools/netbench.py
# for one network spec; per-kernel per-launch means go to stdout.  usage: tools/pmc_netbench.sh <spec> <outdir-tag>
#   e.g. tools/pmc_netbench.sh 5x64x6:f16x3 h3_5x64x6
set -e
spec=$1; tag=$2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -o "$name" -- python3 tools/netbench.py --nets "$spec" > "$out/$name.log" 2>&1 || { tail -5 "$out/$name.log"; return 1; }
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
for p in fetch write tcc mfma sq1 lds; do
  python3 tools/pmc_summary.py "$out/$p" | sed "s|^$out/||"
done
grep -h "ms /" "$out"/fetch.log | tail -1
find "$out" -name "*counter_collection.csv" | head -3 >&2
rm -rf "$out"   # raw rocprofv3 output is large: only the summary travels back
