mkdir -p gpurun_out/r3c
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline > gpurun_out/r3c/$tag.json 2> gpurun_out/r3c/err_$tag.txt || tail -5 gpurun_out/r3c/err_$tag.txt; }
run p4 VNM_DENSE_RING_PAIRS=4
run p4b VNM_DENSE_RING_PAIRS=4
run p2 VNM_DENSE_RING_PAIRS=2
run p1 VNM_DENSE_RING_PAIRS=1
run p4_q4 VNM_DENSE_RING_PAIRS2=4
run p4_q1 VNM_DENSE_RING_PAIRS2=1
run p4_cap64 VNM_DENSE_RING_CAP=64
run p4_nt0 VNM_DENSE_NT=0
run p4_g1 VNM_DENSE_GRID1_PER_CU=1
run old VNM_DENSE_RING=0
run p4c VNM_DENSE_RING_PAIRS=4
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(j['ms_per_step'],3), j['roofline']['kernels_ms'], j['config']['result_rows'])
    except Exception as e: print(f,'ERR',e)
PY
timeout 900 python -m pytest tests/test_gpu_agg.py -x -q -m gpu -k "dense or hot or random" 2>&1 | tail -5
