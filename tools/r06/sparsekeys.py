"""The headline query over SPARSE keys (G distinct random 62-bit values): the dense code range does not apply -- which route, how long.
usage: python tools/r06/sparsekeys.py [rows] [G]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
G = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
g = torch.Generator(device="cuda"); g.manual_seed(5)
vals = torch.randint(0, 1 << 62, (G,), generator=g, device="cuda", dtype=torch.int64)
k = vals[torch.randint(0, G, (n,), generator=g, device="cuda", dtype=torch.int64)]
del vals
v = torch.randint(0, 1 << 14, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.float64) / 128.0
kc, vc = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
lib = L.lib()
names = [b"agg_estimate", b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final", b"agg_finalize", b"agg_part_agg", b"agg_part_scatter", b"agg_hash"]
for rep in range(4):
    if rep == 3: lib.vnm_set_profiling(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
    agg.set_predicate(">", 63.9921875)
    agg.next([kc], [vc, vc], pred=vc, nrows=n)
    cols = agg.result_device()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    ng = agg.result_rows
    agg.close()
sp = {}
for nm in names:
    t, c = ctypes.c_double(0), ctypes.c_int64(0)
    lib.vnm_profile_query(nm, ctypes.byref(t), ctypes.byref(c))
    if c.value: sp[nm.decode()] = round(t.value, 2)
print(sp)
buf = ctypes.create_string_buffer(4000); lib.vnm_route_counts(buf, 4000)
print(f"sparse keys G={G}: {ms:.2f} ms, {ng} groups; routes: {buf.value.decode().replace(chr(10), ' ')[:600]}", flush=True)
