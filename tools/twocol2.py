"""SELECT k, sum(a), sum(b) GROUP BY k over two float64 columns (the dense path's two-value entries; VNM_AGG_NO_DENSE_TWO=1: the hash
partitions' wide entries).  usage: twocol2.py N G"""
import ctypes, sys, time
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2]))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g)
a = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
b = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 64.0
ck, ca, cb = DeviceColumn.from_torch(k), DeviceColumn.from_torch(a), DeviceColumn.from_torch(b)
for rep in range(3):
    L.lib().vnm_set_profiling(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.SUM, 2, pa.float64())])
    agg.next([ck], [ca, cb], nrows=n)
    ng = agg.finish()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    spans = {}
    for nm in (b"agg_estimate", b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final"):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        L.lib().vnm_profile_query(nm, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value:
            spans[nm.decode()[4:]] = (round(ms.value, 2), cnt.value)
    L.lib().vnm_set_profiling(0)
    del agg
print(f"two columns, N={n:.0e} G={G:.0e}: {dt*1e3:.1f} ms, {ng} groups {spans}")
