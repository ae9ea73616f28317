# PMC passes over the DCN window kernels at the bench layer (tools/bench_dcn.py: 256 -> 256 3x3 DCNv2 on 8 x 256 x 256, bf16 operands):
# what the forward's time outside its K-steps waits on.   bash tools/pmc_dcn.sh <tag>  -> profiles/<tag>_dcn_pmc.txt
# Counters in their own runs, with --kernel-trace only (MI355X_MICROARCH.md: rocprofv3 PMC slots).
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r06}
mkdir -p gpurun_out profiles
: > profiles/${TAG}_dcn_pmc.txt
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_dcn_$i -- python3 tools/bench_dcn.py > gpurun_out/pmc_dcn_$i.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_dcn_$i dcn_ 2>&1 >> profiles/${TAG}_dcn_pmc.txt
  find gpurun_out/pmc_dcn_$i -name "*_kernel_trace.csv" -delete
done
cat profiles/${TAG}_dcn_pmc.txt | cut -c1-600
