#!/bin/bash
# round 6, GPU session 10: generic keys below the C ABI (raw ctypes + through the shim), packed-key fused result, pairs default
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "generic or parquet or narrow" > $O/pytest_generic.txt 2>&1
tail -5 $O/pytest_generic.txt | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_vinum_lib.py tests/test_gpu_planner.py tests/test_gpu_csv.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/pytest_shim.txt 2>&1
tail -5 $O/pytest_shim.txt | cut -c1-300
{
  echo "twokeys 5e8 1e6 fused"; python tools/twokeys.py 5e8 1e6 2>&1 | tail -1
  echo "twokeys 5e8 1e8 fused"; python tools/twokeys.py 5e8 1e8 2>&1 | tail -1
  echo "twokeys 5e8 1e6 not fused"; VNM_AGG_NO_PACKED_FUSION=1 python tools/twokeys.py 5e8 1e6 2>&1 | tail -1
  echo "twokeys 5e8 1e8 not fused"; VNM_AGG_NO_PACKED_FUSION=1 python tools/twokeys.py 5e8 1e8 2>&1 | tail -1
} > $O/twokeys3.txt 2>&1
cat $O/twokeys3.txt
timeout 1200 python -m pytest tests/test_gpu_agg.py tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q -m gpu -k "multi or key or packed or random" > $O/pytest_keys.txt 2>&1
tail -4 $O/pytest_keys.txt | cut -c1-300
python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['kernels_ms'], j['check']['ok'])"
