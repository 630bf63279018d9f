"""Nullable KEY column at large G: SELECT k, sum(v), avg(v), count(*) WHERE v > X GROUP BY k with ~12 % NULL keys (and, third line,
NULLs in v as well)."""
import ctypes, sys, time
sys.path.insert(0, ".")
import numpy as np, torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2]))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g)
v = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
def bitmap():
    b = torch.randint(0, 256, ((n + 7) // 8,), device=dev, dtype=torch.uint8, generator=g)
    for _ in range(2):
        b |= torch.randint(0, 256, ((n + 7) // 8,), device=dev, dtype=torch.uint8, generator=g)
    return b
kb, vb = bitmap(), bitmap()
ck = DeviceColumn.from_torch(k)
cv = DeviceColumn.from_torch(v)
ckn = DeviceColumn(k.data_ptr(), kb.data_ptr(), 0, n, pa.int64(), keep=(k, kb))
cvn = DeviceColumn(v.data_ptr(), vb.data_ptr(), 0, n, pa.float64(), keep=(v, vb))
for name, kc, vc in (("no nulls", ck, cv), ("12% NULL keys", ckn, cv), ("12% NULL keys, 12% NULL values", ckn, cvn)):
    for rep in range(2):
        L.lib().vnm_set_profiling(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64()), (L.COUNT_STAR, None, None)], expected_groups=G)
        agg.set_predicate(">", 63.9921875)
        agg.next([kc], [vc, vc, None], pred=vc, nrows=n)
        if len(sys.argv) > 3 and sys.argv[3] == "cols":      # the bench's step: result columns (fused into the final pass where it can be)
            cols = agg.result_device(); ng = agg.result_rows; del cols
        else:
            ng = agg.finish()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        spans = {}
        for nm in (b"agg_pack_keys", b"agg_estimate", b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final", b"agg_part_merge", b"agg_run_patch", b"agg_side_append", b"agg_finalize"):
            ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
            L.lib().vnm_profile_query(nm, ctypes.byref(ms), ctypes.byref(cnt))
            if cnt.value:
                spans[nm.decode()[4:]] = (round(ms.value, 2), cnt.value)
        L.lib().vnm_set_profiling(0)
        del agg
    print(f"{name}: {dt*1e3:.1f} ms, {ng} groups {spans}")
