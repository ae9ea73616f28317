# Every rocprofv3 / bench pass whose summary profiles/README.md indexes for a round, in one gpurun call; the summaries are
# written under gpurun_out/ (merged back by gpurun) AND copied to gpurun_out/profiles_<tag>/ under their profiles/ names:
#   bash tools/prof_round.sh r05      (GPU box)      then here:  cp gpurun_out/profiles_r05/* profiles/
# Counters (--pmc) run in their own passes beside --kernel-trace only (MI355X_MICROARCH.md; gpurun refuses other mixes).
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r06}
OUT=gpurun_out/profiles_$TAG
mkdir -p gpurun_out $OUT
stats() { find "$1" -name "*_kernel_stats.csv" | head -1; }

# 1. the default bench line, plain (no profiler) — what the driver runs
timeout 1500 python3 bench.py > gpurun_out/bench_default_$TAG.log 2>&1
grep "^{\"metric\"" gpurun_out/bench_default_$TAG.log | tail -1 > $OUT/${TAG}_bench_default.json

# 2. bench under rocprofv3: kernel stats, then the two HBM-traffic counter passes; inference and DCN layer likewise
bash tools/prof_all.sh $TAG > gpurun_out/prof_all_$TAG.out 2>&1
head -60 "$(stats gpurun_out/prof_$TAG)" > $OUT/${TAG}_bench_kernel_stats.csv
cp gpurun_out/bench_prof_$TAG.json $OUT/${TAG}_bench_under_rocprof.json
cp gpurun_out/traffic_$TAG.json $OUT/${TAG}_traffic_pmc.json
cp gpurun_out/infer_$TAG.json $OUT/${TAG}_infer_bench.json
head -40 "$(stats gpurun_out/prof_infer_$TAG)" > $OUT/${TAG}_infer_kernel_stats.csv
cp gpurun_out/traffic_infer_$TAG.json $OUT/${TAG}_infer_traffic_pmc.json
tail -1 gpurun_out/dcn_$TAG.json > $OUT/${TAG}_dcn_bench.json
head -30 "$(stats gpurun_out/prof_dcn_$TAG)" > $OUT/${TAG}_dcn_kernel_stats.csv
cp gpurun_out/traffic_dcn_$TAG.json $OUT/${TAG}_dcn_traffic_pmc.json

# 3. config 4 (bf16 convolutions; with the DCN heads and plain): kernel stats of three steps each
bash tools/prof_config4.sh $TAG > gpurun_out/prof_config4_$TAG.out 2>&1
cp profiles/${TAG}_config4_bf16_* $OUT/

# 4. counter passes over the conv16 kernels at the dominant layer
bash tools/pmc_conv16.sh $TAG > gpurun_out/pmc_conv16_$TAG.out 2>&1
cp profiles/${TAG}_conv16_pmc.txt $OUT/

# 4b. counter passes over the DCN window kernels at the bench layer
bash tools/pmc_dcn.sh $TAG > gpurun_out/pmc_dcn_$TAG.out 2>&1
cp profiles/${TAG}_dcn_pmc.txt $OUT/

# 5. one-stream kernel trace of the headline step, broken down by kernel family
RR_WGRAD_STREAM=0 timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_${TAG}_one -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing --no-host-fed > gpurun_out/prof_${TAG}_one.log 2>&1
python3 tools/step_breakdown.py gpurun_out/prof_${TAG}_one > $OUT/${TAG}_step_breakdown_one_stream.txt 2>&1
rm -rf gpurun_out/prof_${TAG}_one

# 6. Soft-NMS sweep, conv16 layer table
timeout 900 python3 tools/bench_softnms.py > gpurun_out/softnms_$TAG.log 2>&1
tail -1 gpurun_out/softnms_$TAG.log > $OUT/${TAG}_softnms_sweep.json
timeout 600 python3 tools/bench_conv16.py > $OUT/${TAG}_conv16_layers.txt 2>&1
ls -la $OUT
