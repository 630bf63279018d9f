"""A stream of 59 HBM-resident 2^24-row batches into one operator, C float64 input columns: SELECT k, sum(c1..cC), count(*) GROUP BY k,
result columns included; synchronous next() per batch against stream mode.  usage: streamcols.py G C [reps]"""
import sys, time, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
B = 1 << 24; nb = 30; n = nb * B
G = int(float(sys.argv[1])); C = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g = torch.Generator(device="cuda"); g.manual_seed(3)
k = torch.randint(0, G, (n,), device="cuda", dtype=torch.int64, generator=g)
cols = [torch.randint(0, 1 << 14, (n,), device="cuda", dtype=torch.int64, generator=g).to(torch.float64) / 128.0 for _ in range(C)]
parts = [(DeviceColumn.from_torch(k[i * B:(i + 1) * B]), [DeviceColumn.from_torch(c[i * B:(i + 1) * B]) for c in cols]) for i in range(nb)]
spans = [b"agg_estimate", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final", b"agg_scan", b"agg_split_join"]
for mode in (0, 1):
    for rep in range(reps):
        L.lib().vnm_set_profiling(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1 + i, pa.float64()) for i in range(C)] + [(L.COUNT_STAR, None, None)], stream_mode=bool(mode))
        for kc, cc in parts:
            a.next([kc], cc + [None], nrows=B)
        out = a.result_device()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        sp = {}
        for nm in spans:
            ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
            L.lib().vnm_profile_query(nm, ctypes.byref(ms), ctypes.byref(cnt))
            if cnt.value:
                sp[nm.decode()] = (round(ms.value, 2), cnt.value)
        L.lib().vnm_set_profiling(0)
        print(f"G={G:.0e} C={C} stream_mode={mode}: {nb} x 2^24 rows: {dt * 1e3:.2f} ms, {a.result_rows} groups; {sp}")
        del a, out
