mkdir -p gpurun_out/r3b
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline > gpurun_out/r3b/$tag.json 2> gpurun_out/r3b/err_$tag.txt || tail -5 gpurun_out/r3b/err_$tag.txt; }
run b512 VNM_DENSE_RING_BLOCK=512
run b512_g3 VNM_DENSE_RING_BLOCK=512 VNM_DENSE_GRID1_PER_CU=3
run b256_g3 VNM_DENSE_RING_BLOCK=256 VNM_DENSE_GRID1_PER_CU=3
run b256_g4 VNM_DENSE_RING_BLOCK=256 VNM_DENSE_GRID1_PER_CU=4 VNM_DENSE_RING_LDS=36
run b256_g2 VNM_DENSE_RING_BLOCK=256 VNM_DENSE_GRID1_PER_CU=2
run b1024 VNM_DENSE_RING_BLOCK=1024
run b512_cap32 VNM_DENSE_RING_BLOCK=512 VNM_DENSE_RING_CAP=32
run b512_nt0 VNM_DENSE_RING_BLOCK=512 VNM_DENSE_NT=0
run old VNM_DENSE_RING=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3b/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(j['ms_per_step'],3), j['roofline']['kernels_ms'], j['config']['result_rows'])
    except Exception as e: print(f,'ERR',e)
PY
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/rp_$c
timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/rp_$c -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --steps 2 --warmup 1 > /tmp/rp_$c.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/rp_$c -name '*.db' | head -1) vnm > $GRAFT_REPO_ROOT/gpurun_out/r3b/pmc_$c.txt
done
cat $GRAFT_REPO_ROOT/gpurun_out/r3b/pmc_*.txt
