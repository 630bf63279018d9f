timeout 1500 python -m pytest tests/test_gpu_agg.py tests/test_gpu_vinum_lib.py tests/test_gpu_float.py tests/test_gpu_bench_check.py tests/test_gpu_planner.py -x -q -m gpu 2>&1 | tail -6
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; tail -3 gpurun_out/bench_b.err
python - <<PY
import json
j=json.loads(open("gpurun_out/bench_b.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms", round(j["ms_per_step"],3), "frac", round(j["roofline"]["frac"],4), j["roofline"]["kernels_ms"])
for k,v in j.get("also",{}).items():
    if k.startswith("configs[3]"): print(k, round(v["ms_per_step"],3), v["roofline"]["kernels_ms"], round(v["roofline"]["frac"],3), v.get("ms_per_batch"))
PY
