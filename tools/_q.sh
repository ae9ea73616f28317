python -m pytest tests/test_bf16_model_gpu.py tests/test_dcn_gpu.py -m gpu -q -x 2>&1 | tail -1
python tools/bench_config4.py --plain --bf16 --steps 8 2>&1 | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.readline());print('c4 plain',d['value'],d['ms_per_step'])"
python tools/bench_config4.py --plain --bf16 --steps 8 2>&1 | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.readline());print('c4 plain',d['value'],d['ms_per_step'])"
python tools/bench_config4.py --steps 8 2>&1 | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.readline());print('c4 dcn',d['value'],d['ms_per_step'])"
