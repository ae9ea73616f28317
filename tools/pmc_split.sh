# PMC passes over the split-operand kernels at the dominant layer (tools/one_layer_split.py): matrix-pipe busy, instruction mix,
# LDS, HBM traffic.   bash tools/pmc_split.sh <tag>   (GPU box; summary -> profiles/<tag>_split_pmc.txt)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r04}
mkdir -p gpurun_out profiles
: > profiles/${TAG}_split_pmc.txt
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_split_$i -- python3 tools/one_layer_split.py > gpurun_out/pmc_split_$i.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_split_$i conv_ 2>&1 | grep "bf16_kernel" >> profiles/${TAG}_split_pmc.txt
done
cat profiles/${TAG}_split_pmc.txt | cut -c1-400
