#!/bin/bash
mkdir -p gpurun_out/r06
for G in 7 1e3 3e3 6e3 1e4 3e4 1e5 3e5 1e6 2e6 4e6 1e7 3e7 1e8 3e8; do
  timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --groups $G --steps 6 --warmup 2 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('G=$G', round(j['ms_per_step'],3), j['roofline'].get('kernels_ms'))"
done | tee gpurun_out/r06/r06_sweep_groups.txt
