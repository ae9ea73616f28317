# rocprofv3 kernel statistics of the headline train step on the split-operand convolutions (cfg.Model.conv_math = f16x3).
#   bash tools/prof_split.sh <tag>      (GPU box, via gpurun; summary -> profiles/<tag>_config2_f16x3_*)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r04}
mkdir -p gpurun_out profiles
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_split_$TAG -- python3 tools/bench_config4.py --plain --fp32 --math f16x3 --steps 3 > gpurun_out/split_bench_$TAG.json 2> gpurun_out/split_bench_$TAG.err
f=$(find gpurun_out/prof_split_$TAG -name "*_kernel_stats.csv" | head -1)
head -45 "$f" > profiles/${TAG}_config2_f16x3_kernel_stats.csv
tail -1 gpurun_out/split_bench_$TAG.json > profiles/${TAG}_config2_f16x3_bench_under_rocprof.json
find gpurun_out/prof_split_$TAG -name "*_kernel_trace.csv" -delete
head -16 profiles/${TAG}_config2_f16x3_kernel_stats.csv | cut -c1-170
