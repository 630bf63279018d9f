"""Skewed keys: k = floor(G * u^p), u uniform -- p = 1 uniform, larger p = heavier head.  Hot-shape query."""
import sys, time
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
for p in (1.0, 2.0, 4.0, 8.0):
    u = torch.rand(n, device=dev, dtype=torch.float64, generator=g)
    k = (u.pow_(p) * G).to(torch.int64); del u
    ck = DeviceColumn.from_torch(k)
    for hint in (G, 0):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())], expected_groups=hint)
            agg.set_predicate(">", 63.9921875)
            agg.next([ck], [cv, cv], pred=cv, nrows=n)
            ng = agg.finish()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            del agg
        print(f"p={p} hint={hint}: {dt*1e3:.1f} ms, {ng} groups")
    del ck, k
