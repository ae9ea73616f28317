cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_split -- python3 tools/bench_config4.py --plain --fp32 --math f16x3 --steps 3 > gpurun_out/split_bench.json 2> gpurun_out/split_bench.err
f=$(find gpurun_out/prof_split -name "*_kernel_stats.csv" | head -1)
head -24 "$f" | cut -c1-220
find gpurun_out/prof_split -name "*_kernel_trace.csv" -delete
tail -1 gpurun_out/split_bench.json
