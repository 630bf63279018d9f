"""Hint below the partitioning threshold, several batches: the first runs the scan kernel, the fill teaches the rest."""
import sys, time
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2])); nb = int(sys.argv[3])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g)
v = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
ck, cv = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64()), (L.COUNT_STAR, None, None)], expected_groups=1000)
for b in range(nb):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg.next([ck], [cv, cv, None], nrows=n)
    torch.cuda.synchronize(); print(f"batch {b}: {(time.perf_counter()-t0)*1e3:.1f} ms")
t0 = time.perf_counter(); ng = agg.finish(); torch.cuda.synchronize()
print(f"finish: {(time.perf_counter()-t0)*1e3:.1f} ms, {ng} groups")
