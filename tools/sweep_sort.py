"""Sort / top-K / filter over sizes, K and column types (a look for cliffs).  usage: python tools/sweep_sort.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
g = torch.Generator(device="cuda"); g.manual_seed(2)

def best(f, reps=3):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b * 1e3

n = 500_000_000
v64 = torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
cols = {"f64": v64, "f32": v64.to(torch.float32), "i64": (v64 * 1e6).to(torch.int64), "i32": (v64 * 1e6).to(torch.int32)}
print("== ORDER BY v DESC LIMIT K (5e8 rows): ms")
for name, t in cols.items():
    c = DeviceColumn.from_torch(t)
    print(name, " ".join(f"K={k}: {best(lambda: ops.sort_indices([c], [1], limit=k)):.2f}" for k in (1, 10, 1000, 100_000, 1_000_000, 10_000_000)), flush=True)
print("== full ORDER BY v (row ids), ms per n")
for name, t in cols.items():
    for m in (10_000_000, 100_000_000, 500_000_000):
        c = DeviceColumn.from_torch(t[:m])
        print(name, f"n={m:.0e}: {best(lambda: ops.sort_indices([c], [0])):.2f}", flush=True)
print("== WHERE v > x -> compacted column (5e8 rows), ms at selectivity 0.01 / 0.5 / 0.99")
for name, t in cols.items():
    c = DeviceColumn.from_torch(t)
    qs = torch.quantile(v64[:10_000_000], torch.tensor([0.99, 0.5, 0.01], device="cuda", dtype=torch.float64)).tolist()
    scale = 1e6 if name.startswith("i") else 1.0
    lit = (lambda q: int(q * scale)) if name.startswith("i") else (lambda q: q)        # (an integer literal against integer columns, as SQL gives it)
    print(name, " ".join(f"{best(lambda: ops.filter_cmp(c, '>', lit(q), [c])):.2f}" for q in qs), flush=True)
