#!/bin/bash
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_round6.py -q -x -m gpu -k "with_nulls" 2>&1 | tail -3
VNM_AGG_TRACE=1 timeout 600 python tools/r06/skew.py 1e9 1e8 2>&1 | grep -v "^$" | cut -c1-300 | head -60
