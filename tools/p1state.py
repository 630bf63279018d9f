"""Which allocation decides the two timing states of the dense path's pass 1 (tuning_log_r02.md: ~5.05 vs ~6.0 ms per process)?
One process, headline query (G = 1e8), pass-1 span measured per phase; between phases either the INPUT columns or the library's
cached region buffers are released and re-allocated, so their placement in HBM changes while everything else stays."""
import ctypes, sys
sys.path.insert(0, ".")
import pyarrow as pa
import torch
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn, pool_trim

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
groups = 100_000_000
dev = torch.device("cuda", 0)
lib = L.lib()


def make_inputs(seed, pad_bytes=0):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    pad = torch.empty(pad_bytes, dtype=torch.uint8, device=dev) if pad_bytes else None
    k = torch.randint(0, groups, (n,), device=dev, dtype=torch.int64, generator=g)
    v = torch.rand(n, device=dev, dtype=torch.float64, generator=g)
    return pad, k, v


def step(k, v):
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
    agg.set_predicate(">", 0.5)
    kc, vc = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
    agg.next([kc], [vc, vc], pred=vc, nrows=n)
    ng = agg.finish()
    agg.close()
    return ng


def measure(tag, k, v, steps=4):
    step(k, v); step(k, v)
    torch.cuda.synchronize()
    lib.vnm_set_profiling(1)
    for _ in range(steps):
        step(k, v)
    torch.cuda.synchronize()
    out = {}
    for nm in (b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final"):
        tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        lib.vnm_profile_query(nm, ctypes.byref(tot), ctypes.byref(cnt))
        out[nm.decode()[9:]] = round(tot.value / max(cnt.value, 1), 3)
    lib.vnm_set_profiling(0)
    print(f"{tag:46s} k@{k.data_ptr():#x} v@{v.data_ptr():#x}  {out}", flush=True)


pad, k, v = make_inputs(1)
measure("initial", k, v)
measure("again (nothing moved)", k, v)
for i, padb in enumerate((0, 2 << 20, 64 << 20, 1 << 30)):
    rel = pool_trim()
    measure(f"regions re-allocated (trim released {rel >> 20} MiB)", k, v)
    del pad, k, v
    torch.cuda.empty_cache()
    pad, k, v = make_inputs(1, padb)
    measure(f"inputs re-allocated after a {padb >> 20} MiB pad", k, v)
