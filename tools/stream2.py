"""The configs[3] one-GPU leg of bench.py on its own (for rocprofv3): 59 HBM-resident batches of 2^24 rows, DENSE keys in [0, G),
hint-less, result columns included.  python tools/stream2.py [groups] [reps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
B = 1 << 24; nb = 59; n = nb * B
groups = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator(device="cuda"); g.manual_seed(3)
k = torch.randint(0, groups, (n,), device="cuda", dtype=torch.int64, generator=g)
v = torch.randint(0, 1 << 14, (n,), device="cuda", dtype=torch.int64, generator=g).to(torch.float64) / 128.0
parts = [(DeviceColumn.from_torch(k[i * B:(i + 1) * B]), DeviceColumn.from_torch(v[i * B:(i + 1) * B])) for i in range(nb)]
for rep in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
    a.set_predicate(">", 63.9921875)
    t_next = []
    for kc, vc in parts:
        t1 = time.perf_counter()
        a.next([kc], [vc, vc], pred=vc, nrows=B)
        t_next.append(time.perf_counter() - t1)
    cols = a.result_device()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t_next.sort()
    print(f"G={groups:.0e}: {nb} batches of 2^24 rows: {dt * 1e3:.2f} ms, {a.result_rows} groups; next(): median {t_next[nb // 2] * 1e6:.0f} us, "
          f"max {t_next[-1] * 1e6:.0f} us, sum {sum(t_next) * 1e3:.2f} ms")
    del a, cols
