#!/bin/bash
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_round6.py -q -x -m gpu -k "sparse_keys" 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 > gpurun_out/r06/bench_check.json 2> gpurun_out/r06/bench_check.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_check.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['frac'], j['check']['ok'])
a=j['also']
for k in a:
    if isinstance(a[k],dict) and 'ms_per_step' in a[k]: print(k, round(a[k]['ms_per_step'],3), a[k]['roofline'].get('kernels_ms') if 'sort' in k else '')
    elif isinstance(a[k],dict) and 'error' in a[k]: print(k, a[k])
PY
tail -3 gpurun_out/r06/bench_check.err
