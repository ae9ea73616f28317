# rocprofv3 kernel stats of the inference post-process bench; summary copied into profiles/ by hand
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out
python3 tools/bench_infer.py --frames 2048 > gpurun_out/infer_$TAG.json 2> gpurun_out/infer_$TAG.err
tail -3 gpurun_out/infer_$TAG.err
cat gpurun_out/infer_$TAG.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_infer_$TAG -- python3 tools/bench_infer.py --frames 1024 --cpu-frames 0 > gpurun_out/infer_prof_$TAG.log 2>&1
tail -2 gpurun_out/infer_prof_$TAG.log
f=$(ls gpurun_out/prof_infer_$TAG/*/*kernel_stats.csv | head -1)
head -25 $f | cut -c1-200
