#!/bin/bash
# round 6, GPU session 6: the rest of the suite after the last failure, new tests, two-key packing, bench default line
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 2400 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_sort_project.py tests/test_gpu_vinum_lib.py tests/test_gpu_planner.py tests/test_gpu_pipeline.py tests/test_gpu_csv.py tests/test_gpu_filter.py tests/test_gpu_float.py -x -q -m gpu > $O/pytest_rest.txt 2>&1
tail -12 $O/pytest_rest.txt
{
for x in 0 1; do
  echo "twokeys 5e8 1e6 dense_fields=$x"; VNM_PACK_DENSE_FIELDS=$x python tools/twokeys.py 5e8 1e6 2>&1 | tail -1
  echo "twokeys 5e8 1e8 dense_fields=$x"; VNM_PACK_DENSE_FIELDS=$x python tools/twokeys.py 5e8 1e8 2>&1 | tail -1
done
} > $O/twokeys.txt 2>&1
cat $O/twokeys.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_default.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernels_ms'], j['roofline']['traffic'])
print('sustained', j.get('sustained')); print('state', j.get('device_state'))
a=j['also']
for k in ("configs[3] one-GPU leg, G=1e6, three input columns","configs[3] one-GPU leg, G=1e6","configs[2] over non-quantised values","configs[0] query shape, large G"):
    print(k, a[k]['ms_per_step'], a[k]['roofline']['kernels_ms'])
PY
