python -m pytest tests/test_dcn_gpu.py -m gpu -q -x 2>&1 | tail -2
python tools/bench_dcn.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ws ', {k:v for k,v in d.items() if 'fwd' in k and 'ms' in k})"
RR_DCN_FWD_WS=0 python tools/bench_dcn.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('old', {k:v for k,v in d.items() if 'fwd' in k and 'ms' in k})"
for e in 7 15; do RR_DCN_EXPF=$e python tools/_qf.py 2>&1 | tail -1; done
