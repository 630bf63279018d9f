#!/bin/bash
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_round6.py -q -x -m gpu -k "string_min_max or generic_keys" 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_vinum_lib.py tests/test_gpu_planner.py tests/test_gpu_csv.py tests/test_gpu_pipeline.py -q -x -m gpu 2>&1 | tail -8
