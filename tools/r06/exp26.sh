#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_round6.py -q -x -m gpu -k "fixed_point" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-also --steps 8 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('default', round(j['ms_per_step'],3), j['roofline']['frac'], j['check']['ok'], j['roofline'].get('kernels_ms'))"
for v in 2 4; do
  VNM_DENSE_FX=0 VNM_DF_EXACT=0 VNM_DENSE_RING_PAIRS2=$v timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --steps 8 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('float64 entries pairs2=$v', round(j['ms_per_step'],3), j['roofline'].get('kernels_ms'))"
done
for v in 4 2; do
  VNM_DENSE_FX=0 VNM_DF_EXACT=0 VNM_DENSE_RING_PAIRS=$v timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --steps 8 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('float64 entries pairs1=$v', round(j['ms_per_step'],3), j['roofline'].get('kernels_ms'))"
done
