#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py -q -x -m gpu 2>&1 | tail -4
timeout 900 python tools/keytypes.py 5e8 2>&1 | grep "^key" | grep -E "int32|int16" | cut -c1-140
