#!/bin/bash
mkdir -p gpurun_out/r06
for v in 2 4 2 4; do
  VNM_DENSE_RING_PAIRS2=$v timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --steps 8 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('pairs2=$v', round(j['ms_per_step'],3), j['roofline'].get('kernels_ms'))"
done
for v in 2 4; do
  VNM_DENSE_RING_PAIRS2=$v timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --groups 1e7 --steps 8 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('G=1e7 pairs2=$v', round(j['ms_per_step'],3), j['roofline'].get('kernels_ms'))"
done
