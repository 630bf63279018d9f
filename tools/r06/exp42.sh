#!/bin/bash
# round 6, GPU session 42: the whole GPU suite, the round's profiles with the final binaries, the default bench line
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 2700 python -m pytest tests -q -m gpu > $O/pytest_all.txt 2>&1
tail -8 $O/pytest_all.txt | cut -c1-300
timeout 2400 bash tools/profile.sh r06 > $O/profile.log 2>&1
tail -3 $O/profile.log
head -8 gpurun_out/prof_r06/r06_rocprofv3_kernel_stats_groupbygroups1e8.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_default.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernels_ms'], j['roofline']['traffic'], j['check']['ok'])
print('sustained', {k:v for k,v in j.get('sustained',{}).items() if k!='device_state_after'})
a=j['also']
for k in a:
    if isinstance(a[k],dict) and 'ms_per_step' in a[k]: print(k, round(a[k]['ms_per_step'],3))
PY
