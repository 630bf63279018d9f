"""The headline query when the rows arrive SORTED by the group key (or in sorted runs): which route, how long.  usage: python tools/r06/sortedkeys.py [rows] [G]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
G = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
g = torch.Generator(device="cuda"); g.manual_seed(5)
k = torch.randint(0, G, (n,), generator=g, device="cuda", dtype=torch.int64)
v = torch.randint(0, 1 << 14, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.float64) / 128.0
lib = L.lib()
for name in ("random", "sorted", "runs of 2^20"):
    if name == "sorted":
        k = torch.sort(k).values
    elif name.startswith("runs"):
        k = k[torch.randperm(n // (1 << 20), device="cuda", generator=g).repeat_interleave(1 << 20) * (1 << 20) + torch.arange(1 << 20, device="cuda").repeat(n // (1 << 20))] if n % (1 << 20) == 0 else k
    kc, vc = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
        agg.set_predicate(">", 63.9921875)
        agg.next([kc], [vc, vc], pred=vc, nrows=n)
        cols = agg.result_device()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        ng = agg.result_rows
        agg.close()
    buf = ctypes.create_string_buffer(300); lib.vnm_route_last(buf, 300)
    print(f"{name}: {ms:.1f} ms, {ng} groups; last route: {buf.value.decode()[:120]}", flush=True)
