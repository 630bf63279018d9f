#!/bin/bash
for v in 3 0 1; do
  echo "== VNM_PART_RING=$v"
  VNM_PART_RING=$v timeout 600 python tools/r06/sparsekeys.py 1e9 1e8 2>&1 | tail -2
  VNM_PART_RING=$v timeout 600 python tools/r06/sparsekeys.py 1e9 1e7 2>&1 | tail -2
  VNM_PART_RING=$v timeout 600 python tools/r06/sparsekeys.py 1e9 1e6 2>&1 | tail -2
done
timeout 2400 python -m pytest tests/test_gpu_agg.py tests/test_gpu_round4.py tests/test_gpu_round5.py -q -x -m gpu 2>&1 | tail -4
