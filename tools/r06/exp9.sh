#!/bin/bash
# round 6, GPU session 9: packed-key fused result columns, pass-1 pairs A/B, remaining tests
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_zz_route_coverage.py -x -q -m gpu > $O/pytest_round6.txt 2>&1
tail -5 $O/pytest_round6.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_agg.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_planner.py tests/test_gpu_vinum_lib.py -x -q -m gpu -k "multi or key or packed or planner or vinum_lib or random" > $O/pytest_keys.txt 2>&1
tail -5 $O/pytest_keys.txt | cut -c1-300
{
for x in 1 0; do
  echo "twokeys 5e8 1e6 no_packed_fusion=$x"; VNM_AGG_NO_PACKED_FUSION=$( [ $x = 1 ] && echo 1 ) python tools/twokeys.py 5e8 1e6 2>&1 | tail -1
  echo "twokeys 5e8 1e8 no_packed_fusion=$x"; VNM_AGG_NO_PACKED_FUSION=$( [ $x = 1 ] && echo 1 ) python tools/twokeys.py 5e8 1e8 2>&1 | tail -1
done
} > $O/twokeys2.txt 2>&1
cat $O/twokeys2.txt
one() { python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms']; print(round(j['ms_per_step'],3), (j.get('check') or {}).get('ok'), ' '.join(f'{n[4:]}={v:.2f}' for n,v in k.items()))"; }
{
for i in 1 2 3; do
  echo -n "pairs=4: "; VNM_DENSE_RING_PAIRS=4 one
  echo -n "pairs=2: "; VNM_DENSE_RING_PAIRS=2 one
done
for g in 1e6 1e7; do for x in 4 2; do echo -n "G=$g pairs=$x: "; VNM_DENSE_RING_PAIRS=$x one --groups $g; done; done
echo -n "stream G=1e6 pairs=4: "; VNM_DENSE_RING_PAIRS=4 one --workload stream --groups 1e6
echo -n "stream G=1e6 pairs=2: "; VNM_DENSE_RING_PAIRS=2 one --workload stream --groups 1e6
} > $O/pairs_ab.txt 2>&1
cat $O/pairs_ab.txt
