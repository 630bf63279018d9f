#!/bin/bash
# N consecutive PROCESSES of the headline bench (the timing of a pass used to depend on where a process's buffers landed), then the
# forced one-rank exchange self-tests with --check; results under gpurun_out/repeat_<tag>/ (copy what is to be kept to profiles/).
TAG=${1:-r03}; N=${2:-5}
OUT=gpurun_out/repeat_$TAG; mkdir -p $OUT
for i in $(seq 1 $N); do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline > $OUT/headline_$i.json 2>/dev/null
done
VNM_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-also --no-cpu-baseline --check > $OUT/exchange_dense_1e8.json 2>/dev/null
VNM_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-also --no-cpu-baseline --check --groups 1e6 > $OUT/exchange_allgather_1e6.json 2>/dev/null
python - <<PY
import json, glob
rows = []
for f in sorted(glob.glob("$OUT/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    rows.append({"file": f.split("/")[-1], "ms_per_step": round(j["ms_per_step"], 3), "frac": round(j["roofline"]["frac"], 4),
                 "kernels_ms": j["roofline"]["kernels_ms"], "exchange_ms_per_step": j.get("exchange_ms_per_step"), "check": j.get("check")})
    print(rows[-1])
json.dump(rows, open("$OUT/summary.json", "w"), indent=1)
PY
