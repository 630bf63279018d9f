timeout 1500 python -m pytest tests/test_gpu_bench_check.py tests/test_gpu_agg.py -x -q -m gpu -k "bench_check or dense_table or partition_aligned" 2>&1 | tail -15
