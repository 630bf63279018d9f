"""Multi-column GROUP BY (MultiNumericalHashAggregate shape) timing: SELECT k1,k2,sum(v),count(*) GROUP BY k1,k2.
usage: multikey.py N G1 G2 [wide]"""
import sys, time
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
g1 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
g2 = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k1 = torch.randint(0, g1, (n,), device=dev, dtype=torch.int64, generator=g)
k2 = torch.randint(0, g2, (n,), device=dev, dtype=torch.int64, generator=g)
if len(sys.argv) > 4 and sys.argv[4] == "wide":      # ranges whose product does not fit 63 bits: the wide-key table
    k1 = k1 * (1 << 44) - (1 << 61)
    k2 = k2 * (1 << 40) + 12345
v = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
c1, c2, cv = DeviceColumn.from_torch(k1), DeviceColumn.from_torch(k2), DeviceColumn.from_torch(v)
import ctypes
lib = L.lib()
for rep in range(3):
    lib.vnm_set_profiling(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.MULTI_NUMERICAL, [pa.int64(), pa.int64()], [(L.SUM, 2, pa.float64()), (L.COUNT_STAR, None, None)],
                              expected_groups=g1 * g2)
    agg.next([c1, c2], [cv, None], nrows=n)
    ng = agg.finish()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    spans = {}
    for nm in (b"agg_pack_keys", b"agg_estimate", b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final"):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        lib.vnm_profile_query(nm, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value:
            spans[nm.decode()] = round(ms.value, 2)
    lib.vnm_set_profiling(0)
    print(f"rep {rep}: {dt*1e3:.1f} ms, {ng} groups, {n/dt/1e9:.2f} Grows/s  {spans}")
