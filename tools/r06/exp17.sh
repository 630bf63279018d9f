#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round4.py tests/test_gpu_round5.py -q -x -m gpu 2>&1 | tail -6
for lean in 1 0; do
  echo "== VNM_DFX_LEAN_SIDE=$lean"
  VNM_DFX_LEAN_SIDE=$lean timeout 600 python tools/r06/skew.py 1e9 1e8 2>&1 | grep "skew G" | tail -1
done
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06/bench_side.json
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_side.json').read())
print(j['ms_per_step'], j['roofline']['frac'], j.get('check',{}).get('ok'))
for k,v in j['also'].items():
    if isinstance(v,dict) and 'ms_per_step' in v and ('NULL' in k or 'skew' in k): print(k, round(v['ms_per_step'],2), v.get('roofline',{}).get('kernels_ms'))
PY
