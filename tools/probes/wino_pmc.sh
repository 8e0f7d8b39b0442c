export TMPDIR=/tmp; export OTH_WINO=1
out=gpurun_out/pmc_wino; rm -rf $out; mkdir -p $out
for pass in "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 tools/netbench.py --nets 10x128x8:f16x3 > $out/$name.log 2>&1
  python3 tools/pmc_summary.py $out/$name | sed "s|^$out/||"
done
rm -rf $out
