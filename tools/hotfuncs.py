"""One f64 input column, any function list: python tools/hotfuncs.py 1e9 1000 sum,count_star  (kernel-level timing via VNM events)."""
import sys, time
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2]))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g)
dt = sys.argv[4] if len(sys.argv) > 4 else "f64"
a = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g)
if dt == "f64":
    a = a.to(torch.float64) / 128.0
ck, ca = DeviceColumn.from_torch(k), DeviceColumn.from_torch(a)
F = {"sum": L.SUM, "min": L.MIN, "max": L.MAX, "count": L.COUNT, "avg": L.AVG, "count_star": L.COUNT_STAR}
for fl in sys.argv[3].split("/"):
    funcs = [(F[f], None if f == "count_star" else 1, None if f == "count_star" else (pa.float64() if dt == "f64" else pa.int64())) for f in fl.split(",")]
    best = 1e9
    for rep in range(3):
        agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], funcs, expected_groups=G)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        agg.next([ck], [ca if f[1] else None for f in funcs], nrows=n)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        ng = agg.finish()
        del agg
    print(f"G={G} {dt} {fl}: {best*1e3:.2f} ms ({ng} groups)")
