mkdir -p gpurun_out/r3d
timeout 1200 python -m pytest tests/test_gpu_agg.py tests/test_gpu_vinum_lib.py tests/test_gpu_float.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/err.txt || tail -20 gpurun_out/r3d/err.txt
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3d/bench.json').read().strip().splitlines()[-1])
print(round(j['ms_per_step'],3), j['roofline']['kernels_ms'], j['roofline']['frac'])
for k,v in j.get('also',{}).items():
    if isinstance(v,dict) and 'ms_per_step' in v: print(k, round(v['ms_per_step'],3), v['roofline']['kernels_ms'], round(v['roofline']['frac'],3))
    elif isinstance(v,dict) and 'by_groups' in v:
        for g,e in v['by_groups'].items(): print('  sweep',g,e)
    else: print(k, v)
PY
