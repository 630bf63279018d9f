"""The hot shape over NON-quantised values (v ~ N(64, 20): nearly every add has a rounding error, so the compensation word of a sum is
touched by nearly every row) against the bench's quantised values (k / 128: exact adds).  usage: realvals.py N G"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2]))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g)
vq = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
vr = torch.randn(n, device=dev, dtype=torch.float64, generator=g) * 20.0 + 64.0
lib = L.lib()
for name, v in (("quantised", vq), ("real", vr)):
    ck, cv = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
    for rep in range(3):
        lib.vnm_set_profiling(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
        agg.set_predicate(">", 64.0)
        agg.next([ck], [cv, cv], pred=cv, nrows=n)
        cols = agg.result_device()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        spans = {}
        for nm in (b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final", b"agg_estimate"):
            ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
            lib.vnm_profile_query(nm, ctypes.byref(ms), ctypes.byref(cnt))
            if cnt.value:
                spans[nm.decode()] = round(ms.value, 2)
        lib.vnm_set_profiling(0)
        del agg, cols
    print(f"{name} G={G}: {dt*1e3:.2f} ms {spans}")
