python -m pytest tests/test_softnms_gpu.py tests/test_infer_gpu.py -m gpu -q -x 2>&1 | tail -3
python tools/bench_softnms.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for r in d['rows']:
    print(r['N'], r['segments'], 'ms', r['ms'], 'us/step', r['us_per_outer_step'], r.get('path'), 'cpu ms', r.get('cpu_oracle_ms_per_segment'))"
