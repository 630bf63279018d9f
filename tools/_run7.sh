for i in 1 2; do
VNM_SORT_TRACE=1 timeout 300 python bench.py --workload topk --limit 0 --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/sorterr.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],2), j['roofline']['kernels_ms'])"
tail -2 gpurun_out/sorterr.txt
done
VNM_SSORT_MIN_ROWS=1000 timeout 900 python -m pytest tests/test_gpu_sort_project.py -x -q -m gpu -k "sample_sort" 2>&1 | tail -5
