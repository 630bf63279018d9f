"""Generic filter shape: WHERE v > X with payload columns (k, v) -> two compacted columns."""
import sys, time, ctypes
sys.path.insert(0, ".")
import torch
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.randint(0, 10**8, (n,), device=dev, dtype=torch.int64, generator=g)
v = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
kc, vc = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
lib = L.lib()
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs, cnt = ops.filter_cmp(vc, ">", 63.9921875, [kc, vc])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"rep {rep}: {dt*1e3:.2f} ms, {cnt} rows, {(16*n + 16*cnt)/dt/1e9:.0f} GB/s algorithmic")
    del outs
