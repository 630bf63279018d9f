"""Single narrow / nullable key at large G: python tools/narrowkey.py 1e9 1e6 int32|nullable  (VNM_AGG_NO_PACK=1 for the old path)."""
import sys, time
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2])); mode = sys.argv[3]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
v = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
if mode == "int32":
    k = torch.randint(0, G, (n,), device=dev, dtype=torch.int32, generator=g)
    ck, kt = DeviceColumn.from_torch(k), pa.int32()
else:
    k = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g)
    bits = torch.randint(0, 256, ((n + 7) // 8,), device=dev, dtype=torch.uint8, generator=g) | torch.randint(0, 256, ((n + 7) // 8,), device=dev, dtype=torch.uint8, generator=g)
    ck, kt = DeviceColumn(k.data_ptr(), bits.data_ptr(), 0, n, pa.int64(), keep=(k, bits)), pa.int64()
cv = DeviceColumn.from_torch(v)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [kt], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64()), (L.COUNT_STAR, None, None)], expected_groups=G)
    agg.next([ck], [cv, cv, None], nrows=n)
    ng = agg.finish()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    del agg
print(f"{mode} key, G={G}: {dt*1e3:.1f} ms, {ng} groups")
