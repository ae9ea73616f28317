export TMPDIR=/tmp
TAG=r02
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_dcn_${c}_$TAG -- python3 tools/bench_dcn.py > gpurun_out/pmc_dcn_${c}_$TAG.log 2>&1
done
python3 tools/pmc_traffic.py gpurun_out/pmc_dcn_FETCH_SIZE_$TAG gpurun_out/pmc_dcn_WRITE_SIZE_$TAG > gpurun_out/traffic_dcn_$TAG.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dcn_$TAG -- python3 tools/bench_dcn.py > gpurun_out/dcn_$TAG.json 2> gpurun_out/dcn_$TAG.err
# config 4 at model level: kernel stats of three train steps with DCN heads, and the same loop without them
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_config4_$TAG -- python3 tools/bench_config4.py --steps 3 > gpurun_out/config4_$TAG.json 2> gpurun_out/config4_$TAG.err
timeout 600 python3 tools/bench_config4.py --steps 3 --plain > gpurun_out/config4_plain_$TAG.json 2>> gpurun_out/config4_$TAG.err
