# rocprofv3 kernel statistics of the config-4 train step on bf16 convolutions: with the DCN heads, and plain.
#   bash tools/prof_config4.sh <tag>      (GPU box, via gpurun; summaries -> profiles/<tag>_config4_bf16_*)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r04}
mkdir -p gpurun_out profiles
for mode in dcn plain; do
  extra=""; [ "$mode" = plain ] && extra="--plain --bf16"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4${mode}_$TAG -- python3 tools/bench_config4.py --steps 3 $extra > gpurun_out/config4_bf16_${mode}_$TAG.json 2> gpurun_out/config4_bf16_${mode}_$TAG.err
  f=$(find gpurun_out/prof_c4${mode}_$TAG -name "*_kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/${TAG}_config4_bf16_${mode}_kernel_stats_full.csv
  cp "$f" profiles/${TAG}_config4_bf16_${mode}_kernel_stats.csv          # untruncated
  tail -1 gpurun_out/config4_bf16_${mode}_$TAG.json > profiles/${TAG}_config4_bf16_${mode}_bench.json
  find gpurun_out/prof_c4${mode}_$TAG -name "*_kernel_trace.csv" -delete
done
head -30 profiles/${TAG}_config4_bf16_plain_kernel_stats.csv | cut -c1-200
