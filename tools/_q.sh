python -m pytest tests/test_streams_gpu.py -m gpu -q -x -k "bf16 or dcn" 2>&1 | tail -3
python -m pytest tests/test_conv_bf16_gpu.py tests/test_bf16_model_gpu.py tests/test_dcn_gpu.py -m gpu -q -x 2>&1 | tail -2
