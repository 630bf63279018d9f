"""One key holds a large share of the rows (here: NULL keys, `frac` of all rows) while the rest spread over G groups.
usage: heavykey.py N G frac program   (program: count | sum | sum2)"""
import ctypes, sys, time
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])); G = int(float(sys.argv[2])); frac = float(sys.argv[3]); prog = sys.argv[4]
mode = sys.argv[5] if len(sys.argv) > 5 else "null"      # null: the heavy key is NULL; value: the heavy key is the value 7
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g) * 977 - 5
heavy = torch.rand(n, device=dev, generator=g) < frac
validity = None
if mode == "null":
    pad = (-n) % 64
    bits = torch.cat([~heavy, torch.ones(pad, dtype=torch.bool, device=dev)]).view(-1, 8).to(torch.uint8)
    w = (bits * torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)).sum(1).to(torch.uint8).contiguous()
    validity = w
else:
    k = torch.where(heavy, torch.full_like(k, 7), k)
v = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
b = torch.randint(-2**40, 2**40, (n,), device=dev, dtype=torch.int64, generator=g)
ck = DeviceColumn.from_torch(k, validity=validity)
cv, cb = DeviceColumn.from_torch(v), DeviceColumn.from_torch(b)
spec, inputs = {"count": ([(L.COUNT_STAR, None, None)], [None]),
                "sum": ([(L.SUM, 1, pa.float64())], [cv]),
                "sum2": ([(L.SUM, 1, pa.float64()), (L.MAX, 2, pa.int64())], [cv, cb])}[prog]
lib = L.lib()
for rep in range(3):
    lib.vnm_set_profiling(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], spec)
    agg.next([ck], inputs, nrows=n)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ng = agg.finish()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    phases = f"create+next {1e3*(t1-t0):.1f} ms, finish {1e3*(time.perf_counter()-t1):.1f} ms"

    spans = {}
    for nm in (b"agg_pack_keys", b"agg_estimate", b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final", b"agg_part_merge", b"agg_demote"):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        lib.vnm_profile_query(nm, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value:
            spans[nm.decode()[4:]] = (round(ms.value, 2), cnt.value)
    lib.vnm_set_profiling(0)
    agg.close()
print(f"{mode} G={G} frac={frac} {prog}: {dt*1e3:.1f} ms ({phases}), {ng} groups  {spans}")
