#!/bin/bash
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_round6.py -q -x -m gpu -k "entry_word" 2>&1 | tail -3
for v in "VNM_XSORT_READY_LIST=1" "VNM_XSORT_READY_LIST=0" "VNM_XSORT_READY_LIST=1" "VNM_XSORT_READY_LIST=0"; do
  echo "== $v"
  env $v timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --workload topk --limit 0 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline'].get('kernels_ms'))"
done
