#!/bin/bash
for v in "VNM_SSORT_PAIRS1=2" "VNM_SSORT_PAIRS2=2" "VNM_SSORT_PAIRS1=2 VNM_SSORT_PAIRS2=2" "VNM_SORT_APX=0"; do
  echo "== $v"
  env VNM_SORT_APX=0 $v timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --workload topk --limit 0 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline'].get('kernels_ms'))"
done
