"""More shapes over G (a look for cliffs): the hot query with ~12 % NULLs in the value column / in the key column, and with a two-column key.
usage: python tools/sweep_nulls_keys.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
g = torch.Generator(device="cuda"); g.manual_seed(2)
v = torch.randint(0, 1 << 14, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.float64) / 128.0
bits = torch.randint(0, 256, ((n + 7) // 8,), device="cuda", dtype=torch.uint8, generator=g) | torch.randint(0, 256, ((n + 7) // 8,), device="cuda", dtype=torch.uint8, generator=g) | \
       torch.randint(0, 256, ((n + 7) // 8,), device="cuda", dtype=torch.uint8, generator=g)          # ~12.5 % zero bits
for shape in ("plain", "null_values", "null_keys", "two_keys"):
    for groups in (7, 1000, 10_000, 100_000, 1_000_000, 2_000_000, 10_000_000, 100_000_000):
        k = torch.randint(0, groups, (n,), generator=g, device="cuda", dtype=torch.int64)
        kc = DeviceColumn.from_torch(k, validity=bits if shape == "null_keys" else None)
        vc = DeviceColumn.from_torch(v, validity=bits if shape == "null_values" else None)
        keys, ktypes, kind = [kc], [pa.int64()], L.SINGLE_NUMERICAL
        if shape == "two_keys":
            k2 = (k % 16).contiguous(); k1 = (k // 16).contiguous()
            keys, ktypes, kind = [DeviceColumn.from_torch(k1), DeviceColumn.from_torch(k2)], [pa.int64(), pa.int64()], L.MULTI_NUMERICAL
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            agg = ops.DeviceAggregate(kind, ktypes, [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
            agg.set_predicate(">", 63.9921875)
            agg.next(keys, [vc, vc], pred=vc, nrows=n)
            cols = agg.result_device()
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            agg.close()
        print(f"{shape:12s} G={groups:<10d} {best * 1e3:8.2f} ms per {n:.1e} rows", flush=True)
        del k, kc
