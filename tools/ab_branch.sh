run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r05_ab_$name.json 2> gpurun_out/r05_ab_$name.err
  python -c "
import json;d=json.load(open('gpurun_out/r05_ab_$name.json'));print('$name', d['value'], d['ms_per_step'], d['step_mfma_frac'], d['roofline']['frac'], d['host_enqueue_ms_per_step'], d['collectives_per_step'])"
}
run b0_q8 RR_BRANCH_STREAMS=0 GPU_MAX_HW_QUEUES=8
run b3_q8 RR_BRANCH_STREAMS=3 GPU_MAX_HW_QUEUES=8
run b5_q8 RR_BRANCH_STREAMS=5 GPU_MAX_HW_QUEUES=8
run dp_b0 RR_BRANCH_STREAMS=0 RR_DP_FORCE=1
run dp_b3 RR_BRANCH_STREAMS=3 RR_DP_FORCE=1
run dp_b3_q8 RR_BRANCH_STREAMS=3 RR_DP_FORCE=1 GPU_MAX_HW_QUEUES=8
run dp_b5_q8 RR_BRANCH_STREAMS=5 RR_DP_FORCE=1 GPU_MAX_HW_QUEUES=8
