#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_sort_project.py tests/test_gpu_round6.py -q -x -m gpu -k "sort or order" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "order_by" 2>&1 | tail -3
for apx in 1 0; do
  echo "== VNM_SORT_APX=$apx"
  VNM_SORT_APX=$apx timeout 600 python bench.py --no-cpu-baseline --no-also --workload topk --limit 0 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j.get('check'), j['roofline'].get('kernels_ms'))"
done
