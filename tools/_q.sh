python -m pytest tests -m gpu -q -x --deselect "tests/test_configs_gpu.py::test_config2_hourglass104_train_mode_vs_oracle" --durations=12 -k "not streams" > gpurun_out/r05_gputest_c.log 2>&1; tail -22 gpurun_out/r05_gputest_c.log
timeout 300 python tools/bench_conv16.py 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['shape'], {k:v for k,v in d.items() if 'tflops' in k or 'err' in k})"
python tools/bench_config4.py --plain --bf16 --steps 3 > /dev/null 2>&1
python tools/bench_config4.py --plain --bf16 --steps 8 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print('c4 plain',d['value'],d['ms_per_step'],d['allocator']['allocated_peak_GiB'])"
python tools/bench_config4.py --steps 8 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print('c4 dcn',d['value'],d['ms_per_step'],d['allocator']['allocated_peak_GiB'])"
