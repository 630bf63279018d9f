"""No GROUP BY (OneGroupAggregate shape): SELECT sum(v), avg(v), min(v), count(*) [WHERE v > X]."""
import sys, time
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
v = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
cv = DeviceColumn.from_torch(v)
for pred in (False, True):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        agg = ops.DeviceAggregate(L.ONE_GROUP, [], [(L.SUM, 0, pa.float64()), (L.AVG, 0, pa.float64()), (L.MIN, 0, pa.float64()), (L.COUNT_STAR, None, None)])
        if pred:
            agg.set_predicate(">", 63.9921875)
        agg.next([], [cv, cv, cv, None], pred=cv if pred else None, nrows=n)
        ng = agg.finish()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"pred={pred}: {dt*1e3:.2f} ms, {ng} group, {8*n/dt/1e9:.0f} GB/s")
