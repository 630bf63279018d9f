#!/bin/bash
run() { timeout 150 python bench.py --no-cpu-baseline --no-also "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.3f kernels_ms %.3f frac %.3f rows %d %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['config']['result_rows'], r.get('kernels_ms')))"; }
for g in 7 30000 1000000 10000000 100000000; do echo "== groupby NO HINT G=$g"; run --workload groupby --groups $g --steps 5 --warmup 2 --no-hint; done
