#!/bin/bash
mkdir -p gpurun_out/r06
for v in 15 16; do
for G in 1e4 2e4 3e4 6e4 1e5; do
  VNM_DENSE_FX_SMALL_MIN_BITS=$v timeout 600 python bench.py --no-cpu-baseline --no-also --groups $G --steps 6 --warmup 2 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('minbits=$v G=$G', round(j['ms_per_step'],3), j['check']['ok'], j['roofline'].get('kernels_ms'))"
done
done
timeout 600 python bench.py --no-cpu-baseline --no-also --workload stream --groups 3e4 --steps 6 --warmup 2 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('stream G=3e4', round(j['ms_per_step'],3), j['check']['ok'], j['roofline'].get('kernels_ms'))"
timeout 1500 python -m pytest tests/test_gpu_round6.py -q -x -m gpu -k "fixed_point" 2>&1 | tail -3
