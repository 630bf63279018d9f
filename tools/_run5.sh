mkdir -p gpurun_out/r3e
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-also --no-cpu-baseline --check $EXTRA > gpurun_out/r3e/$tag.json 2> gpurun_out/r3e/err_$tag.txt || tail -12 gpurun_out/r3e/err_$tag.txt; }
run dense VNM_BENCH_FORCE_EXCHANGE=1
run dense2 VNM_BENCH_FORCE_EXCHANGE=1
EXTRA="--groups 1e7" run dense_1e7 VNM_BENCH_FORCE_EXCHANGE=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e/dense*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(j['ms_per_step'],3), j['roofline']['kernels_ms'], j.get('exchange_ms_per_step'), j.get('check'))
    except Exception as e: print(f,'ERR',e)
PY
