"""GROUP BY three wide int64 columns with G distinct values EACH (beyond 63 bits even as dictionary codes: the tuple dictionary),
SUM(float64): kernel spans per run.  usage: widekey.py N G"""
import sys, time, ctypes
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2]))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
f64 = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
base = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g)
col = DeviceColumn.from_torch
ks = [col(base * 977_000_003 - (1 << 61)), col((base ^ 0x5DEECE66D) * 1_000_003 + 17), col(base * (1 << 33) - 99)]
lib = L.lib()
for rep in range(3):
    lib.vnm_set_profiling(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.MULTI_NUMERICAL, [pa.int64()] * 3, [(L.SUM, 1, pa.float64())])
    agg.next(ks, [col(f64)], nrows=n)
    ng = agg.finish()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    spans = {}
    for nm in (b"agg_estimate", b"agg_tuple_ids", b"agg_tuple_keys", b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final", b"agg_pack_keys"):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        lib.vnm_profile_query(nm, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value: spans[nm.decode()] = (round(ms.value, 2), cnt.value)
    lib.vnm_set_profiling(0)
    print(f"G={G}: {dt*1e3:.1f} ms, {ng} groups {spans}", flush=True)
    agg.close()
