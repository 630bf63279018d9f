"""ORDER BY v DESC, a LIMIT K over 1e9 rows: candidates selected on the first key, only they are sorted by both keys."""
import sys, time
sys.path.insert(0, ".")
import torch
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(2)
v = torch.randn(n, device=dev, dtype=torch.float64, generator=g) * 3.0 + 11.0
a = torch.randint(0, 1 << 40, (n,), device=dev, dtype=torch.int64, generator=g)
cv, ca = DeviceColumn.from_torch(v), DeviceColumn.from_torch(a)
for k in (10, 1000, 100000):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx = ops.sort_indices([cv, ca], [L.DESC, L.ASC], limit=k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"ORDER BY v DESC, a LIMIT {k}: {dt*1e3:.2f} ms")
