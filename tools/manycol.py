"""C input columns: SELECT k, sum(c1), ..., sum(cC), count(*) GROUP BY k  (wide partition entries carry key + C values).
usage: manycol.py N G C [int]"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2])); C = int(sys.argv[3])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g)
INT = len(sys.argv) > 4 and sys.argv[4] == "int"     # int64 columns (128-bit sums) instead of float64
cols = [torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g) if INT else torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0 for _ in range(C)]
ck = DeviceColumn.from_torch(k)
cc = [DeviceColumn.from_torch(c) for c in cols]
if len(sys.argv) > 4 and sys.argv[-1] == "null":    # ~12 % NULLs in every input column
    bits = [torch.randint(0, 256, ((n + 7) // 8,), device=dev, dtype=torch.uint8, generator=g) | torch.randint(0, 256, ((n + 7) // 8,), device=dev, dtype=torch.uint8, generator=g) | torch.randint(0, 256, ((n + 7) // 8,), device=dev, dtype=torch.uint8, generator=g) for _ in range(C)]
    cc = [DeviceColumn(c.values_ptr, b.data_ptr(), 0, n, c.arrow_type if hasattr(c, "arrow_type") else pa.float64(), keep=(c, b)) for c, b in zip(cc, bits)]
import ctypes
lib = L.lib()
for rep in range(3):
    lib.vnm_set_profiling(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1 + i, pa.int64() if INT else pa.float64()) for i in range(C)] + [(L.COUNT_STAR, None, None)],
                              expected_groups=G)
    agg.next([ck], cc + [None], nrows=n)
    ng = agg.finish()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    spans = {}
    for nm in (b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final", b"agg_split_join", b"agg_split_finish", b"agg_split_sort", b"agg_split_units", b"agg_estimate"):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        lib.vnm_profile_query(nm, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value:
            spans[nm.decode()] = round(ms.value, 2)
    lib.vnm_set_profiling(0)
    print(f"C={C} G={G}: {dt*1e3:.1f} ms, {ng} groups  {spans}")
    del agg
