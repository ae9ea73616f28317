export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
RR_WGRAD_STREAM=0 timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r05_one -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/prof_r05_one.log 2>&1
python3 tools/step_breakdown.py gpurun_out/prof_r05_one > gpurun_out/r05_step_breakdown_one_stream.txt 2>&1
rm -rf gpurun_out/prof_r05_one
head -120 gpurun_out/r05_step_breakdown_one_stream.txt
