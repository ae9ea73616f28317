python -m pytest tests/test_dcn_gpu.py -m gpu -q -x 2>&1 | tail -2
for e in 0 2 16 0; do RR_DCN_EXP=$e python tools/_q.py 2>&1 | tail -1; done
RR_DCN_DGRAD_DMA=0 python tools/_q.py 2>&1 | tail -1
