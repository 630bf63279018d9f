import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1]))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.zeros(n, device=dev, dtype=torch.int64)
v = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
kb = torch.zeros((n + 7) // 8, device=dev, dtype=torch.uint8)
vb = torch.randint(0, 256, ((n + 7) // 8,), device=dev, dtype=torch.uint8, generator=g)
ckn = DeviceColumn(k.data_ptr(), kb.data_ptr(), 0, n, pa.int64(), keep=(k, kb))
cv = DeviceColumn.from_torch(v)
cvn = DeviceColumn(v.data_ptr(), vb.data_ptr(), 0, n, pa.float64(), keep=(v, vb))
for name, vc in (("all-NULL keys", cv), ("all-NULL keys, nullable v", cvn)):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64()), (L.COUNT_STAR, None, None)], expected_groups=1000000)
        agg.next([ckn], [vc, vc, None], nrows=n)
        ng = agg.finish()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        del agg
    print(f"{name}: {dt*1e3:.2f} ms for {n} rows, {ng} groups")
