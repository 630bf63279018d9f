"""Skewed keys: k = floor(G * u^p), u uniform -- p = 1 uniform, larger p = heavier head.  Hot-shape query."""
import ctypes, sys, time
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
G = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**7
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
v = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
cv = DeviceColumn.from_torch(v)
ps = [float(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1.0, 2.0, 4.0, 8.0]
for p in ps:
    u = torch.rand(n, device=dev, dtype=torch.float64, generator=g)
    k = (u.pow_(p) * G).to(torch.int64); del u
    ck = DeviceColumn.from_torch(k)
    for hint in ((G, 0) if len(sys.argv) <= 4 else (0,)):
        for rep in range(2):
            L.lib().vnm_set_profiling(1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())], expected_groups=hint)
            agg.set_predicate(">", 63.9921875)
            agg.next([ck], [cv, cv], pred=cv, nrows=n)
            if len(sys.argv) > 5 and sys.argv[5] == "cols":     # the bench's step: the result columns (fused into the final pass)
                cols = agg.result_device(); ng = agg.result_rows; del cols
            else:
                ng = agg.finish()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            spans = {}
            for nm in (b"agg_estimate", b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final", b"agg_part_merge", b"agg_run_patch", b"agg_side_append", b"agg_finalize"):
                ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
                L.lib().vnm_profile_query(nm, ctypes.byref(ms), ctypes.byref(cnt))
                if cnt.value:
                    spans[nm.decode()[4:]] = (round(ms.value, 2), cnt.value)
            L.lib().vnm_set_profiling(0)
            del agg
        print(f"p={p} hint={hint}: {dt*1e3:.1f} ms, {ng} groups {spans}")
    del ck, k
