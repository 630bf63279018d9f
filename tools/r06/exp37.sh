#!/bin/bash
for v in "" "VNM_DENSE_RING_LDS=64" "VNM_DENSE_RING_LDS=64 VNM_DENSE_GRID1_PER_CU=4" "VNM_DENSE_RING_LDS=72" "" "VNM_DENSE_RING_LDS=64"; do
  env $v timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --steps 8 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('G=1e8 [$v]', round(j['ms_per_step'],3), j['roofline'].get('kernels_ms'))"
done
for v in "" "VNM_DENSE_RING_LDS=64"; do
  env $v timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --groups 1e6 --steps 8 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('G=1e6 [$v]', round(j['ms_per_step'],3), j['roofline'].get('kernels_ms'))"
done
