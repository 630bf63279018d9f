"""sum(expr) inside the aggregate: fused in-register evaluation vs the reference's plan (project, then aggregate).
Run on the GPU box:  python tools/exprbench.py [groups]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn

n = int(1e9)
groups = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e8)
g = torch.Generator(device="cuda"); g.manual_seed(3)
k = torch.randint(0, groups, (n,), device="cuda", dtype=torch.int64, generator=g)
mk = lambda hi: torch.randint(0, hi, (n,), device="cuda", dtype=torch.int64, generator=g).to(torch.float64) / 128.0
total, tax, tip = mk(128), mk(64), mk(32)
cols = {"total": DeviceColumn.from_torch(total), "tax": DeviceColumn.from_torch(tax), "tip": DeviceColumn.from_torch(tip)}
kc = DeviceColumn.from_torch(k)
expr = ("mul", ("mul", ("sub", 1, "total"), ("add", 2, "tax")), ("sub", 1, "tip"))
lib = L.lib()

def fused():
    a = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 7, pa.float64()), (L.COUNT_STAR, None, None)])
    a.set_input_expr(0, expr, list(cols))
    a.next([kc], [None, None], nrows=n, expr_cols=list(cols.values()))
    return a.finish()

def planned():   # the reference's plan: Project(keep input) -> Aggregate
    e = ops.project(expr, cols, length=n)
    a = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 7, pa.float64()), (L.COUNT_STAR, None, None)])
    a.next([kc], [e, None], nrows=n)
    return a.finish()

for name, fn in (("fused (expression evaluated in registers)", fused), ("project, then aggregate (the reference's plan)", planned)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ng = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"G={groups:.0e} {name}: {ms:.2f} ms per 1e9 rows, {ng} groups; algorithmic {32 * n / ms / 1e6:.0f} GB/s (4 columns read once)")
