#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round4.py -q -x -m gpu -k "columns or column or multi" 2>&1 | tail -3
for v in 1 0 3 1 0; do
  echo "== VNM_FXN_READY_LIST=$v"
  VNM_FXN_READY_LIST=$v python tools/manycol.py 5e8 1e8 3 2>&1 | tail -1
  VNM_FXN_READY_LIST=$v python tools/manycol.py 5e8 1e6 3 2>&1 | tail -1
done
