python -m pytest tests/test_dcn_gpu.py tests/test_bf16_model_gpu.py -m gpu -q -x 2>&1 | tail -2
python -m pytest tests/test_streams_gpu.py -m gpu -q -x -k "dcn" 2>&1 | tail -2
python tools/bench_dcn.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print({k:v for k,v in d.items() if 'ms' in k})"
