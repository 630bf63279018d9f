run() { env "$@" VNM_SORT_TRACE=0 timeout 300 python bench.py --workload topk --limit 0 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms']; print('$*', round(j['ms_per_step'],2), {x:round(k[x],2) for x in k if x.startswith('sort_')})"; }
run A=0
run VNM_SSORT_PAIRS1=2
run VNM_SSORT_PAIRS2=2
run VNM_SSORT_PAIRS1=2 VNM_SSORT_PAIRS2=2
run VNM_SSORT_GRID1_PER_CU=2
run VNM_SSORT_SPLIT2=2
