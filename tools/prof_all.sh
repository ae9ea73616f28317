# All rocprofv3 passes whose summaries go to profiles/ (run on the GPU box via gpurun): bench kernel stats + PMC traffic,
# inference kernel stats + PMC traffic, DCN layer kernel stats + PMC traffic.   bash tools/prof_all.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out
bash tools/prof_bench.sh $TAG > gpurun_out/prof_bench_$TAG.out 2>&1
bash tools/prof_infer.sh $TAG > gpurun_out/prof_infer_$TAG.out 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_infer_${c}_$TAG -- python3 tools/bench_infer.py --frames 256 --cpu-frames 0 > gpurun_out/pmc_infer_${c}_$TAG.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_dcn_${c}_$TAG -- python3 tools/bench_dcn.py > gpurun_out/pmc_dcn_${c}_$TAG.log 2>&1
done
python3 tools/pmc_traffic.py gpurun_out/pmc_infer_FETCH_SIZE_$TAG gpurun_out/pmc_infer_WRITE_SIZE_$TAG > gpurun_out/traffic_infer_$TAG.json
python3 tools/pmc_traffic.py gpurun_out/pmc_dcn_FETCH_SIZE_$TAG gpurun_out/pmc_dcn_WRITE_SIZE_$TAG > gpurun_out/traffic_dcn_$TAG.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dcn_$TAG -- python3 tools/bench_dcn.py > gpurun_out/dcn_$TAG.json 2> gpurun_out/dcn_$TAG.err
ls gpurun_out | grep $TAG | head -60
# the raw traces are tens of MB each (gpurun merges at most 64 MiB back): keep the summaries only
# (this tag's directories only: tools/trace_gaps.py and a re-run of tools/pmc_traffic.py read other tags' raw traces)
for d in gpurun_out/*_$TAG; do
  [ -d "$d" ] || continue
  find "$d" -name "*_kernel_trace.csv" -delete
  find "$d" -name "*_counter_collection.csv" -delete
done
du -sh gpurun_out
