#!/bin/bash
# round 6, GPU session 2: fixed-point entry words (VNM_DENSE_FX) A/B + geometry knobs, the new parity tests, then the whole GPU suite
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $O/pytest_round6.txt 2>&1
tail -15 $O/pytest_round6.txt
one() { python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms']; print(round(j['ms_per_step'],3), (j.get('check') or {}).get('ok'), ' '.join(f'{n[4:]}={v:.2f}' for n,v in k.items()))"; }
{
for i in 1 2; do
  echo -n "G=1e8 fx=0: "; VNM_DENSE_FX=0 one
  echo -n "G=1e8 fx=1: "; VNM_DENSE_FX=1 one
done
echo -n "G=1e8 fx=1 tb13: "; VNM_DENSE_TBITS=13 one
echo -n "G=1e8 fx=1 p1=8: "; VNM_DENSE_P1=8 one
echo -n "G=1e8 fx=1 p1=6: "; VNM_DENSE_P1=6 one
echo -n "G=1e8 fx=1 pairs2=4: "; VNM_DENSE_RING_PAIRS2=4 one
echo -n "G=1e8 fx=1 tb13 p1=7 pairs2=4: "; VNM_DENSE_TBITS=13 VNM_DENSE_RING_PAIRS2=4 one
echo -n "G=1e8 fx=1 nt=0: "; VNM_DENSE_NT=0 one
echo -n "G=1e8 fx=1 nt=3: "; VNM_DENSE_NT=3 one
for g in 1e6 1e7; do for x in 0 1; do echo -n "G=$g fx=$x: "; VNM_DENSE_FX=$x one --groups $g; done; done
} > $O/fx_ab.txt 2>&1
cat $O/fx_ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -8 $O/pytest_all.txt
