#!/bin/bash
# ms per step over the group count for the three program shapes of bench.py (hint-less, 1e9 rows): a look for cliffs
for shape in hot minmax count_star; do
  for g in 1e3 3e3 1e4 3e4 1e5 2e5 5e5 1e6 2e6 5e6 1e7 2e7 5e7 1e8; do
    python bench.py --shape $shape --groups $g --no-cpu-baseline --no-also --no-check --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$shape', '$g', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['roofline']['kernels_ms'].items()})"
  done
done
