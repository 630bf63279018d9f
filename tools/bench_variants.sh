#!/bin/bash
run() { timeout 90 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.3f kernel_ms %.3f GB/s %.0f frac %.3f rows %d' % (d['ms_per_step'], r['kernel_ms'], r['achieved'], r['frac'], d['config']['result_rows']))"; }
echo "== filter default"; run --workload filter --steps 10
echo "== filter 1024 x2"; VNM_FILTER_WGS_PER_CU=2 run --workload filter --steps 10
echo "== filter 512"; VNM_FILTER_THREADS=512 run --workload filter --steps 10
echo "== filter s=0.01"; run --workload filter --steps 10 --selectivity 0.01
echo "== filter s=0.99"; run --workload filter --steps 10 --selectivity 0.99
