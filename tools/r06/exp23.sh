#!/bin/bash
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_agg.py -q -x -m gpu 2>&1 | tail -4
for v in 3 0 3 0; do
  for G in 1e8 1e6; do
    VNM_DENSE_READY_LIST=$v timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --groups $G --steps 8 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('list=$v G=$G', round(j['ms_per_step'],3), j['roofline'].get('kernels_ms'))"
  done
done
VNM_DENSE_READY_LIST=3 timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --workload stream --steps 8 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('stream list=3', round(j['ms_per_step'],3), j['roofline'].get('kernels_ms'))"
VNM_DENSE_READY_LIST=0 timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --workload stream --steps 8 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('stream list=0', round(j['ms_per_step'],3), j['roofline'].get('kernels_ms'))"
