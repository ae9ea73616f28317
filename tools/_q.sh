python -m pytest tests/test_dcn_gpu.py -m gpu -q 2>&1 | grep -E "^E|FAILED|passed|failed" | head -40
