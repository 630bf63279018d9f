#!/bin/bash
# the entry-word sample sort: parity against the LSD sort, then the full-sort bench line with and without it
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_vinum_lib.py -q -m gpu -k "inexact_values or string_column_as_group_key" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_round6.py -q -x -m gpu -k "entry_word" 2>&1 | tail -15
for apx in 1 0; do
  echo "== VNM_SORT_APX=$apx"
  VNM_SORT_TRACE=1 VNM_SORT_APX=$apx timeout 600 python bench.py --no-cpu-baseline --no-also --workload topk --limit 0 --steps 5 --warmup 2 2>gpurun_out/r06/sort_apx_$apx.err | tail -1 > gpurun_out/r06/sort_apx_$apx.json
  grep "\[sort\]" gpurun_out/r06/sort_apx_$apx.err | sort | uniq -c | head -8
  python - <<PY
import json
j=json.loads(open('gpurun_out/r06/sort_apx_$apx.json').read())
print(j['ms_per_step'], j.get('check'), j['roofline'].get('kernels_ms'))
PY
done
for v in "VNM_XSORT_GRID1_PER_CU=2" "VNM_XSORT_NT=1" "VNM_XSORT_LOCAL_PER_CU=24" "VNM_XSORT_LOCAL_PER_CU=96" "VNM_XSORT_SAMPLE=1048576" "VNM_XSORT_SPLIT2=2"; do
  echo "== $v"
  env $v VNM_SORT_TRACE=1 timeout 600 python bench.py --no-cpu-baseline --no-also --no-check --workload topk --limit 0 --steps 5 --warmup 2 2>gpurun_out/r06/x.err | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline'].get('kernels_ms'))"
  grep "long kernel" gpurun_out/r06/x.err | tail -1
done
