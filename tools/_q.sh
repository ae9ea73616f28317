python -m pytest tests/test_bf16_model_gpu.py -m gpu -q -x -k "every_kernel" 2>&1 | grep -E "Error|error|assert|bad|^E " | head -20
for i in 1 2; do python tools/bench_config4.py --plain --bf16 --steps 8 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print('c4 plain',d['value'],d['ms_per_step'],d['allocator']['allocated_peak_GiB'])"; done
