#!/bin/bash
mkdir -p gpurun_out/r06
for v in 1 0; do
for G in 3e4 6e4 1e5 2e5 3e5 5e5; do
  VNM_DENSE_FX_SMALL=$v timeout 600 python bench.py --no-cpu-baseline --no-also --groups $G --steps 6 --warmup 2 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('small=$v G=$G', round(j['ms_per_step'],3), j['check']['ok'], j['roofline'].get('kernels_ms'))"
done
done
VNM_DENSE_FX_SMALL=1 timeout 600 python bench.py --no-cpu-baseline --no-also --workload stream --groups 1e5 --steps 6 --warmup 2 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('stream G=1e5 small=1', round(j['ms_per_step'],3), j['check']['ok'], j['roofline'].get('kernels_ms'))"
VNM_DENSE_FX_SMALL=0 timeout 600 python bench.py --no-cpu-baseline --no-also --workload stream --groups 1e5 --steps 6 --warmup 2 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('stream G=1e5 small=0', round(j['ms_per_step'],3), j['check']['ok'], j['roofline'].get('kernels_ms'))"
timeout 2400 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round4.py tests/test_gpu_agg.py -q -x -m gpu 2>&1 | tail -4
