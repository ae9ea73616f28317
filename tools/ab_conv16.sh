# A/B of the config-4 train step: csrc/conv16.hip (RR_CONV16=1, default) against the round-4 kernels (RR_CONV16=0)
for v in 1 0; do
  RR_CONV16=$v python tools/bench_config4.py --plain --bf16 --steps 5 > gpurun_out/r05_c4_plain_c16_$v.json 2> gpurun_out/r05_c4_plain_c16_$v.err
  RR_CONV16=$v python tools/bench_config4.py --steps 5 > gpurun_out/r05_c4_dcn_c16_$v.json 2> gpurun_out/r05_c4_dcn_c16_$v.err
  python - <<P
import json
for k in ("plain","dcn"):
    try:
        d=json.load(open("gpurun_out/r05_c4_%s_c16_$v.json"%k)); print("conv16=$v",k,d["value"],d["ms_per_step"],d["allocator"]["allocated_peak_GiB"])
    except Exception as e:
        print("conv16=$v",k,"FAILED",e); print(open("gpurun_out/r05_c4_%s_c16_$v.err"%k).read()[-1500:])
P
done
