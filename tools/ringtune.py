"""Sweep of the ring-scatter switches of the dense group-by path inside ONE process (same input placement for every variant):
headline query (N rows, G = 1e8, s = 0.5), spans per kernel.  usage: python tools/ringtune.py [N] [repeat]"""
import ctypes, os, sys
sys.path.insert(0, ".")
import pyarrow as pa
import torch
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn, pool_trim

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
repeat = int(sys.argv[2]) if len(sys.argv) > 2 else 2
groups = int(float(os.environ.get("GROUPS", "1e8")))
dev = torch.device("cuda", 0)
lib = L.lib()
g = torch.Generator(device=dev); g.manual_seed(1)
j = torch.randint(0, 1 << 14, (n,), device=dev, dtype=torch.int64, generator=g)
v = j.to(torch.float64) / 128.0
del j
k = torch.randint(0, groups, (n,), device=dev, dtype=torch.int64, generator=g)
kc, vc = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)


def step():
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
    agg.set_predicate(">", 63.9921875)
    agg.next([kc], [vc, vc], pred=vc, nrows=n)
    ng = agg.finish()
    agg.close()
    return ng


def measure(tag, env, steps=6):
    keys = [e.split("=")[0] for e in env]
    for e in env:
        a, b = e.split("="); os.environ[a] = b
    step(); step()
    torch.cuda.synchronize()
    lib.vnm_set_profiling(1)
    for _ in range(steps):
        ng = step()
    torch.cuda.synchronize()
    out = {}
    for nm in (b"agg_estimate", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final"):
        tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        lib.vnm_profile_query(nm, ctypes.byref(tot), ctypes.byref(cnt))
        out[nm.decode()[4:]] = round(tot.value / max(cnt.value, 1), 3)
    lib.vnm_set_profiling(0)
    print(f"{tag:34s} sum {sum(out.values()):7.3f}  {out}  groups {ng}", flush=True)
    for a in keys:
        os.environ.pop(a, None)


VARIANTS = [l.split() for l in os.environ.get("VARIANTS", "").split(";") if l.strip()] or [
    ["default"],
    ["p2", "VNM_DENSE_RING_PAIRS=2"],
    ["g1", "VNM_DENSE_GRID1_PER_CU=1"],
    ["p2_g1", "VNM_DENSE_RING_PAIRS=2", "VNM_DENSE_GRID1_PER_CU=1"],
    ["nt0", "VNM_DENSE_NT=0"],
    ["q4", "VNM_DENSE_RING_PAIRS2=4"],
    ["cap64", "VNM_DENSE_RING_CAP=64"],
    ["old", "VNM_DENSE_RING=0"],
]
for r in range(repeat):
    for var in VARIANTS:
        measure(var[0], var[1:])
    if r + 1 < repeat:
        print("-- pool trimmed:", pool_trim() >> 20, "MiB released", flush=True)
