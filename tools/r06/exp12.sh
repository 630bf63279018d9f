#!/bin/bash
# the suite after the last fixes: the round's own tests first, then the shim suites, then everything
mkdir -p gpurun_out/r06
python -m pytest tests/test_cabi.py -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_round6.py -q -x -m gpu 2>&1 | tail -15 | tee gpurun_out/r06/pytest_round6.txt
timeout 1500 python -m pytest tests/test_gpu_vinum_lib.py tests/test_gpu_planner.py tests/test_gpu_csv.py tests/test_gpu_pipeline.py -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06/pytest_shim.txt
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r06/pytest_full.txt
python bench.py --steps 10 --warmup 3 --no-also 2>/dev/null | tail -1 > gpurun_out/r06/bench_after.json
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_after.json').read())
print(j['ms_per_step'], j['roofline']['frac'], j.get('check'))
PY
