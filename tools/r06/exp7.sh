#!/bin/bash
# round 6, GPU session 7: narrow (32-bit) entry words A/B + tests
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $O/pytest_round6.txt 2>&1
tail -12 $O/pytest_round6.txt | cut -c1-300
one() { python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms']; print(round(j['ms_per_step'],3), (j.get('check') or {}).get('ok'), ' '.join(f'{n[4:]}={v:.2f}' for n,v in k.items()))"; }
{
for i in 1 2; do
  echo -n "G=1e8 narrow=0: "; VNM_DENSE_FX_NARROW=0 one
  echo -n "G=1e8 narrow=1: "; VNM_DENSE_FX_NARROW=1 one
done
echo -n "G=1e8 narrow=1 tb13: "; VNM_DENSE_TBITS=13 one
echo -n "G=1e8 narrow=1 p1=7: "; VNM_DENSE_P1=7 one
for g in 1e6 3e6 1e7; do for x in 0 1; do echo -n "G=$g narrow=$x: "; VNM_DENSE_FX_NARROW=$x one --groups $g; done; done
echo -n "stream G=1e6 narrow=0: "; VNM_DENSE_FX_NARROW=0 one --workload stream --groups 1e6
echo -n "stream G=1e6 narrow=1: "; VNM_DENSE_FX_NARROW=1 one --workload stream --groups 1e6
} > $O/narrow_ab.txt 2>&1
cat $O/narrow_ab.txt
timeout 1500 python -m pytest tests/test_gpu_agg.py tests/test_gpu_fullsize.py tests/test_gpu_round4.py tests/test_gpu_bench_check.py -x -q -m gpu > $O/pytest_some.txt 2>&1
tail -6 $O/pytest_some.txt | cut -c1-300
