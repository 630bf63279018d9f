"""ORDER BY v DESC end to end: sort indices + take of two 8-byte columns (Sort::Sorted = SortIndices + Take, sort.cpp:22-40)."""
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
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx = ops.sort_indices([cv], [L.DESC])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sv = ops.take(cv, idx, n)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    sa = ops.take(ca, idx, n)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"rep {rep}: sort {1e3*(t1-t0):.1f} ms, take(v) {1e3*(t2-t1):.1f} ms, take(a) {1e3*(t3-t2):.1f} ms")
    del idx, sv, sa
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, sk = ops.sort_indices_keyed([cv], [L.DESC])       # the sorted key comes out of the last radix pass: no gather for it
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sa = ops.take(ca, idx, n)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep {rep}: keyed sort {1e3*(t1-t0):.1f} ms (sorted key returned: {sk is not None}), take(a) {1e3*(t2-t1):.1f} ms")
    del idx, sk, sa
