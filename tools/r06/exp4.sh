#!/bin/bash
# round 6, GPU session 4: after the flags fix -- lean final pass A/B, one-level geometry, the one-pass multi-column path
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $O/pytest_round6.txt 2>&1
tail -15 $O/pytest_round6.txt
one() { python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms']; print(round(j['ms_per_step'],3), (j.get('check') or {}).get('ok'), ' '.join(f'{n[4:]}={v:.2f}' for n,v in k.items()))"; }
{
for i in 1 2; do
  echo -n "G=1e8 lean=0: "; VNM_DFX_LEAN=0 one
  echo -n "G=1e8 lean=1: "; VNM_DFX_LEAN=1 one
done
echo -n "G=1e8 fx=0: "; VNM_DENSE_FX=0 one
echo -n "G=1e8 lean=1 occ2: "; VNM_PA_OCC=2 one
echo -n "G=1e8 lean=1 tb13: "; VNM_DENSE_TBITS=13 one
echo -n "G=1e8 lean=1 p1=8: "; VNM_DENSE_P1=8 one
for g in 1e6 3e6 1e7; do for x in 0 1; do echo -n "G=$g fx=$x: "; VNM_DENSE_FX=$x one --groups $g; done; done
echo -n "G=1e6 fx=1 one_level=0: "; VNM_DENSE_FX_ONE_LEVEL=0 one --groups 1e6
} > $O/fx_ab3.txt 2>&1
cat $O/fx_ab3.txt
{
for x in 0 1; do
 echo "manycol 5e8 1e8 3 fxn=$x"; VNM_DENSE_FXN=$x python tools/manycol.py 5e8 1e8 3 2>&1 | tail -3
 echo "manycol 5e8 1e6 3 fxn=$x"; VNM_DENSE_FXN=$x python tools/manycol.py 5e8 1e6 3 2>&1 | tail -3
 echo "manycol 5e8 1e7 2 fxn=$x"; VNM_DENSE_FXN=$x python tools/manycol.py 5e8 1e7 2 2>&1 | tail -3
done
} > $O/manycol.txt 2>&1
cat $O/manycol.txt
python bench.py --steps 10 --warmup 3 > $O/bench_full.json 2> $O/bench_full.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_full.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline'])
for k,v in j.get('also',{}).items():
    print(k, '->', v.get('ms'), v.get('roofline_frac'))
PY
timeout 1500 python -m pytest tests/test_gpu_agg.py tests/test_gpu_fullsize.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_bench_check.py -x -q -m gpu > $O/pytest_some.txt 2>&1
tail -8 $O/pytest_some.txt
