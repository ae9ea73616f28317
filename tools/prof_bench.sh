export TMPDIR=/tmp
timeout 1400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_prof_r1b.log 2>&1
tail -1 gpurun_out/bench_prof_r1b.log | cut -c1-200
