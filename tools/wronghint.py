"""Hint too small by a factor: python tools/wronghint.py 1e9 1e8 100  (hot shape, expected_groups = G / factor)."""
import sys, time
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2])); f = float(sys.argv[3])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g)
v = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
ck, cv = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64()), (L.COUNT_STAR, None, None)], expected_groups=max(1, int(G / f)))
    agg.set_predicate(">", 63.9921875)
    agg.next([ck], [cv, cv, None], pred=cv, nrows=n)
    ng = agg.finish()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    del agg
print(f"G={G}, hint={int(G / f)}: {dt*1e3:.1f} ms, {ng} groups")
