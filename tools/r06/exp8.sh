#!/bin/bash
# round 6, GPU session 8: pass-1 knobs with narrow words, then the whole suite
mkdir -p gpurun_out/r06
O=gpurun_out/r06
one() { python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms']; print(round(j['ms_per_step'],3), (j.get('check') or {}).get('ok'), ' '.join(f'{n[4:]}={v:.2f}' for n,v in k.items()))"; }
{
echo -n "default: "; one
echo -n "grid1=1: "; VNM_DENSE_GRID1_PER_CU=1 one
echo -n "grid1=3: "; VNM_DENSE_GRID1_PER_CU=3 one
echo -n "pairs=2: "; VNM_DENSE_RING_PAIRS=2 one
echo -n "nt=0: "; VNM_DENSE_NT=0 one
echo -n "nt=3: "; VNM_DENSE_NT=3 one
echo -n "ring_lds=96: "; VNM_DENSE_RING_LDS=96 one
echo -n "ring_lds=144: "; VNM_DENSE_RING_LDS=144 one
echo -n "p1=9: "; VNM_DENSE_P1=9 one
echo -n "odd_cap=0: "; VNM_DENSE_ODD_CAP=0 one
echo -n "default: "; one
} > $O/pass1_knobs.txt 2>&1
cat $O/pass1_knobs.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -8 $O/pytest_all.txt | cut -c1-300
