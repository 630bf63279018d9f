#!/bin/bash
# round 6, GPU session 1: the L2-atomic table bench, and the exact-add final pass (VNM_DF_EXACT = 0 off / 1 proved / 2 assumed) A/B in alternating processes
mkdir -p gpurun_out/r06
O=gpurun_out/r06
(rocm-smi --showclocks --showpower --showtemp --showmemuse 2>&1 | head -60) > $O/smi_before.txt
timeout 600 tools/l2_atomic_bench > $O/l2_atomic.txt 2>&1
one() { python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms']; print(round(j['ms_per_step'],3), j.get('check'), ' '.join(f'{n[4:]}={v:.2f}' for n,v in k.items()))"; }
for i in 1 2 3; do
  for x in 0 1 2; do echo -n "G=1e8 exact=$x: "; VNM_DF_EXACT=$x one; done
done > $O/exact_ab.txt 2>&1
for x in 0 1; do echo -n "G=1e6 exact=$x: "; VNM_DF_EXACT=$x one --groups 1e6; done >> $O/exact_ab.txt 2>&1
for x in 0 1; do echo -n "G=1e7 exact=$x: "; VNM_DF_EXACT=$x one --groups 1e7; done >> $O/exact_ab.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_float.py tests/test_gpu_agg.py -x -q -m gpu > $O/pytest_subset.txt 2>&1
tail -5 $O/pytest_subset.txt
cat $O/exact_ab.txt
cat $O/l2_atomic.txt
