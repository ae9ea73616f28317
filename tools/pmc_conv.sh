# PMC passes over the dominant-layer conv micro-benchmark (tools/bench_conv.py 0): MFMA busy, LDS, waits
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_conv_$i -- python3 tools/bench_conv.py 0 > gpurun_out/pmc_conv_$i.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_conv_$i conv 2>&1 | tail -4
done
