# PMC passes over the 16-bit-activation kernels at the dominant layer (tools/one_layer_conv16.py): matrix-pipe busy, instruction mix,
# LDS conflicts, HBM traffic.   bash tools/pmc_conv16.sh <tag>   (GPU box; summary -> profiles/<tag>_conv16_pmc.txt)
# Counters in their own runs, with --kernel-trace only (MI355X_MICROARCH.md: rocprofv3 PMC slots; gpurun refuses --pmc beside other traces).
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r05}
mkdir -p gpurun_out profiles
: > profiles/${TAG}_conv16_pmc.txt
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAIT_ANY SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_conv16_$i -- python3 tools/one_layer_conv16.py > gpurun_out/pmc_conv16_$i.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_conv16_$i conv16 2>&1 >> profiles/${TAG}_conv16_pmc.txt
  find gpurun_out/pmc_conv16_$i -name "*_kernel_trace.csv" -delete
done
cat profiles/${TAG}_conv16_pmc.txt | cut -c1-600
