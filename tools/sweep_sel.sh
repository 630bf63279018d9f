#!/bin/bash
# the hot query over the selectivity of its WHERE and the group count (1e9 rows, hint-less): ms per step
for s in 0.01 0.99; do
  for g in 1e4 1e5 1e6 2e6 5e6 1e7 1e8; do
    python bench.py --selectivity $s --groups $g --no-cpu-baseline --no-also --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('s=$s', '$g', round(d['ms_per_step'],2), d['check']['ok'], {k: round(v,2) for k,v in d['roofline']['kernels_ms'].items()})"
  done
done
