"""The configs[3] one-GPU leg of bench.py on its own (for rocprofv3): 59 HBM-resident batches of 2^24 rows, DENSE keys in [0, G),
hint-less, result columns included.  python tools/stream2.py [groups] [reps] [async: 0 | 1]
(async = vnm_agg_set_async: the batches wait in the operator and go to the device as the segments of one launch)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
B = 1 << 24; nb = 59; n = nb * B
groups = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
modes = [int(sys.argv[3])] if len(sys.argv) > 3 else [0, 1]
spans = [b"agg_estimate", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final", b"agg_scan", b"agg_finalize"]
g = torch.Generator(device="cuda"); g.manual_seed(3)
k = torch.randint(0, groups, (n,), device="cuda", dtype=torch.int64, generator=g)
v = torch.randint(0, 1 << 14, (n,), device="cuda", dtype=torch.int64, generator=g).to(torch.float64) / 128.0
parts = [(DeviceColumn.from_torch(k[i * B:(i + 1) * B]), DeviceColumn.from_torch(v[i * B:(i + 1) * B])) for i in range(nb)]
for mode in modes:
  for rep in range(reps):
    L.lib().vnm_set_profiling(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())], stream_mode=bool(mode))
    a.set_predicate(">", 63.9921875)
    t_next = []
    for kc, vc in parts:
        t1 = time.perf_counter()
        a.next([kc], [vc, vc], pred=vc, nrows=B)
        t_next.append(time.perf_counter() - t1)
    cols = a.result_device()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t_next.sort()
    import ctypes
    sp = {}
    for nm in spans:
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        L.lib().vnm_profile_query(nm, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value:
            sp[nm.decode()] = (round(ms.value, 3), cnt.value)
    L.lib().vnm_set_profiling(0)
    print(f"G={groups:.0e} async={mode}: {nb} batches of 2^24 rows: {dt * 1e3:.2f} ms, {a.result_rows} groups; next(): median {t_next[nb // 2] * 1e6:.0f} us, "
          f"max {t_next[-1] * 1e6:.0f} us, sum {sum(t_next) * 1e3:.2f} ms; spans {sp}")
    del a, cols
