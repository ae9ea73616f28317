# rocprofv3 passes of the default bench command; summaries are copied into profiles/ by hand
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out
timeout 1400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/bench_prof_$TAG.log 2>&1
grep "^{\"metric\"" gpurun_out/bench_prof_$TAG.log > gpurun_out/bench_prof_$TAG.json
timeout 1400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_$TAG -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout 1400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_$TAG -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/pmc_write_$TAG.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG > gpurun_out/traffic_$TAG.json
cat gpurun_out/traffic_$TAG.json | head -30
