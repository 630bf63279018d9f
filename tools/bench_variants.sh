#!/bin/bash
run() { timeout 150 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.3f kernels_ms %.3f frac %.3f rows %d %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['config']['result_rows'], r.get('kernels_ms')))"; }
echo "== topk K=10"; run --workload topk --limit 10 --steps 5 --warmup 2
echo "== topk K=1e6"; run --workload topk --limit 1000000 --steps 3 --warmup 1
echo "== project"; run --workload project --steps 5 --warmup 2
for g in 7 1000 100000 10000000 100000000; do echo "== groupby G=$g"; run --workload groupby --groups $g --steps 5 --warmup 2; done
