"""1e9 rows as a stream of 2^24-row batches (what TableReaderOperator hands the aggregate), G groups, hinted.
python tools/stream.py [groups]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(1e9); B = 1 << 24
groups = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000
g = torch.Generator(device="cuda"); g.manual_seed(3)
k = torch.randint(0, groups, (n,), device="cuda", dtype=torch.int64, generator=g) * 1000003
v = torch.randint(0, 1 << 14, (n,), device="cuda", dtype=torch.int64, generator=g).to(torch.float64) / 128.0
def run(hint):
    a = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())], expected_groups=hint)
    a.set_predicate(">", 63.9921875)
    for lo in range(0, n, B):
        hi = min(n, lo + B)
        kc, vc = DeviceColumn.from_torch(k[lo:hi]), DeviceColumn.from_torch(v[lo:hi])
        a.next([kc], [vc, vc], pred=vc, nrows=hi - lo)
    return a.finish()
for hint in (groups, 0):
    run(hint); torch.cuda.synchronize()
    t0 = time.perf_counter(); ng = run(hint); torch.cuda.synchronize()
    print(f"G={groups:.0e} hint={hint}: {n // B + 1} batches of 2^24 rows: {(time.perf_counter() - t0) * 1e3:.1f} ms, {ng} groups")
