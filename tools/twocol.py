"""Two input columns: SELECT k, sum(a), max(b), count(*) GROUP BY k."""
import sys, time
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2]))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g)
a = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
b = torch.randint(-2**40, 2**40, (n,), device=dev, dtype=torch.int64, generator=g)
ck, ca, cb = DeviceColumn.from_torch(k), DeviceColumn.from_torch(a), DeviceColumn.from_torch(b)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.MAX, 2, pa.int64()), (L.COUNT_STAR, None, None)], expected_groups=G)
    agg.next([ck], [ca, cb, None], nrows=n)
    ng = agg.finish()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"G={G}: {dt*1e3:.1f} ms, {ng} groups")
    del agg
