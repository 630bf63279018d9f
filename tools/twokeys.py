"""Two-column key, SELECT k1, k2, sum(v), avg(v) WHERE v > X GROUP BY k1, k2 (k1 = k // 16, k2 = k % 16, k uniform in [0, G)).  usage: twokeys.py N G"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); groups = int(float(sys.argv[2]))
g = torch.Generator(device="cuda"); g.manual_seed(2)
v = torch.randint(0, 1 << 14, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.float64) / 128.0
k = torch.randint(0, groups, (n,), generator=g, device="cuda", dtype=torch.int64)
k2 = (k % 16).contiguous(); k1 = (k // 16).contiguous()
keys = [DeviceColumn.from_torch(k1), DeviceColumn.from_torch(k2)]
vc = DeviceColumn.from_torch(v)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.MULTI_NUMERICAL, [pa.int64(), pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
    agg.set_predicate(">", 63.9921875)
    agg.next(keys, [vc, vc], pred=vc, nrows=n)
    t1 = time.perf_counter()
    cols = agg.result_device()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    agg.close()
print(f"two keys N={n:.1e} G={groups:.1e}: {dt * 1e3:.2f} ms (next {1e3 * (t1 - t0):.2f})")
