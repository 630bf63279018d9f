#!/bin/bash
# fixed-point words with nullable keys / values: parity, then the bench's NULL variants
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_round6.py -q -x -m gpu -k "with_nulls" 2>&1 | tail -12
timeout 1800 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_agg.py -q -x -m gpu 2>&1 | tail -6
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06/bench_nulls.json
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_nulls.json').read())
print(j['ms_per_step'], j['roofline']['frac'], j.get('check',{}).get('ok'))
for k,v in j['also'].items():
    if isinstance(v,dict) and 'ms_per_step' in v: print(k, round(v['ms_per_step'],2), v.get('roofline',{}).get('kernels_ms'))
PY
