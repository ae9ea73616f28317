# HBM traffic of the split-operand kernels at the dominant layer: FETCH_SIZE and WRITE_SIZE in separate passes (tools/pmc_traffic.py
# applies the guide's unit and gfx950 corrections).   bash tools/pmc_split_traffic.sh <tag>  -> profiles/<tag>_split_traffic_pmc.json
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r04}
mkdir -p gpurun_out profiles
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_sfetch_$TAG -- python3 tools/one_layer_split.py > gpurun_out/pmc_sfetch_$TAG.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_swrite_$TAG -- python3 tools/one_layer_split.py > gpurun_out/pmc_swrite_$TAG.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_sfetch_$TAG gpurun_out/pmc_swrite_$TAG > profiles/${TAG}_split_traffic_pmc.json
head -30 profiles/${TAG}_split_traffic_pmc.json
