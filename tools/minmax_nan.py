"""What the ordered mode of float MIN / MAX costs: SELECT k, min(v), max(v) GROUP BY k over N rows, G groups, with a
share of NaN / -0.0 rows (0 = clean data: flag pass only).  usage: python tools/minmax_nan.py [N] [G] [share]"""
import os, sys, time
import torch
import pyarrow as pa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 28
groups = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**6
share = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
g = torch.Generator(device="cuda"); g.manual_seed(1)
k = torch.randint(0, groups, (n,), generator=g, device="cuda", dtype=torch.int64)
v = torch.randint(1, 1 << 14, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.float64) / 128.0
if share > 0:
    u = torch.rand(n, generator=g, device="cuda")
    v = torch.where(u < share / 2, torch.full_like(v, float("nan")), v)
    v = torch.where((u >= share / 2) & (u < share), torch.full_like(v, -0.0), v)
kc, vc = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.MIN, 1, pa.float64()), (L.MAX, 1, pa.float64())])
    agg.next([kc], [vc, vc], nrows=n)
    ng = agg.finish()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    agg.close()
print(f"N={n:.3g} G={groups:.3g} special share {share}: {dt * 1e3:.1f} ms ({ng} groups) = {n / dt / 1e9:.2f} Grows/s")
