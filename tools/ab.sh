#!/bin/bash
# A/B of two builds of the library on ONE box: vinum_amd/lib (new) against vinum_amd/lib_old (e.g. built from a git worktree of the
# previous commit), alternating processes; prints every run and the medians.  usage: bash tools/ab.sh [pairs] [bench.py args]
P=${1:-3}; shift
cp vinum_amd/lib/libvinum_hip.so /tmp/new.so; cp vinum_amd/lib_old/libvinum_hip.so /tmp/old.so
one() { python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms']; print(round(j['ms_per_step'],3), ' '.join(f'{n[4:]}={v:.2f}' for n,v in k.items()))"; }
for i in $(seq 1 $P); do
  cp /tmp/old.so vinum_amd/lib/libvinum_hip.so; echo -n "old: "; one "$@"
  cp /tmp/new.so vinum_amd/lib/libvinum_hip.so; echo -n "new: "; one "$@"
done | tee /tmp/ab.log
python - <<'PY'
import statistics as st
rows = [l.split() for l in open('/tmp/ab.log')]
for tag in ('old:', 'new:'):
    r = [x for x in rows if x[0] == tag]
    print(tag, 'median ms/step', st.median(float(x[1]) for x in r), {k: st.median(float(dict(y.split('=') for y in x[2:])[k]) for x in r) for k in dict(y.split('=') for y in r[0][2:])})
PY
