"""The hot query over key / value columns of other types than int64 / float64 (a look for cliffs): SELECT k, sum(v), avg(v) WHERE v > X GROUP BY k.
usage: python tools/keytypes.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
for ktype, vtype in [(torch.int64, torch.float64), (torch.int32, torch.float64), (torch.int64, torch.float32), (torch.int32, torch.float32), (torch.int64, torch.int64), (torch.int32, torch.int32), (torch.int16, torch.float64)]:
    for groups in (7, 1000, 1_000_000, 100_000_000):
        if ktype == torch.int16 and groups > 30000:
            continue
        k = torch.randint(0, groups, (n,), generator=g, device="cuda", dtype=torch.int64).to(ktype)
        v = torch.randint(0, 1 << 14, (n,), generator=g, device="cuda", dtype=torch.int64)
        v = (v.to(torch.float64) / 128.0).to(vtype) if vtype.is_floating_point else v.to(vtype)
        kc, vc = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
        at = {torch.float64: pa.float64(), torch.float32: pa.float32(), torch.int64: pa.int64(), torch.int32: pa.int32()}[vtype]
        kt = {torch.int64: pa.int64(), torch.int32: pa.int32(), torch.int16: pa.int16()}[ktype]
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [kt], [(L.SUM, 1, at), (L.AVG, 1, at)])
            agg.set_predicate(">", 63.9921875 if vtype.is_floating_point else 8191)
            agg.next([kc], [vc, vc], pred=vc, nrows=n)
            cols = agg.result_device()
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            agg.close()
        import ctypes
        buf = ctypes.create_string_buffer(400); L.lib().vnm_route_last(buf, 400)
        print(f"key {str(ktype)[6:]:6s} value {str(vtype)[6:]:8s} G={groups:<10d} {best * 1e3:8.2f} ms per {n:.1e} rows   last route: {buf.value.decode()[:90]}")
        del k, v, kc, vc
