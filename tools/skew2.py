import sys, time, ctypes
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2])); p = float(sys.argv[3])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
v = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
cv = DeviceColumn.from_torch(v)
u = torch.rand(n, device=dev, dtype=torch.float64, generator=g)
k = (u.pow_(p) * G).to(torch.int64); del u
ck = DeviceColumn.from_torch(k)
lib = L.lib()
for rep in range(2):
    lib.vnm_set_profiling(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())], expected_groups=G)
    agg.set_predicate(">", 63.9921875)
    agg.next([ck], [cv, cv], pred=cv, nrows=n)
    ng = agg.finish()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out = {}
    for nm in [b"agg_estimate", b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final"]:
        ms, c = ctypes.c_double(0), ctypes.c_int64(0)
        lib.vnm_profile_query(nm, ctypes.byref(ms), ctypes.byref(c))
        if c.value: out[nm.decode()] = (round(ms.value, 2), c.value)
    lib.vnm_set_profiling(0)
    print(f"p={p}: {dt*1e3:.1f} ms, {ng} groups, {out}")
    del agg
