"""Cliff finder: one process, a matrix of GROUP BY shapes (key kind x group count x aggregate program x predicate) over the same
resident columns, second-run wall time each, scaled to ms per 1e9 rows and sorted -- anything far above its neighbours took a
slow dispatch path.  usage: cliffs.py [N] [nulls]   (nulls: the second matrix -- nullable keys and nullable input columns)"""
import itertools, sys, time
sys.path.insert(0, ".")
import torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
f64 = torch.randint(0, 2**14, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.float64) / 128.0
i64 = torch.randint(-2**40, 2**40, (n,), device=dev, dtype=torch.int64, generator=g)
i32 = torch.randint(-2**20, 2**20, (n,), device=dev, dtype=torch.int32, generator=g)
f32 = torch.rand(n, device=dev, dtype=torch.float32, generator=g)
f64b, f64c, i64b, i64c = f64 * 3.0 + 1.0, 4096.0 - f64, i64 // 3, -i64
valid_bits = torch.randint(0, 256, ((n + 7) // 8,), device=dev, dtype=torch.uint8, generator=g) | 1   # ~half the rows NULL


NULLS = len(sys.argv) > 2 and sys.argv[2] == "nulls"


def col(t, nullable=False):
    return DeviceColumn.from_torch(t, validity=valid_bits if nullable else None)


def keyset(kind, G):
    base = torch.randint(0, G, (n,), device=dev, dtype=torch.int64, generator=g)
    if kind == "i64":
        return [(col(base * 977 - 5), pa.int64())]
    if kind == "i32":
        return [(col(base.to(torch.int32)), pa.int32())]
    if kind == "f64":
        return [(col(base.to(torch.float64) * 0.25 - 3.0), pa.float64())]
    if kind == "i64x2":
        g1 = max(1, int(G ** 0.5))
        return [(col(base % g1), pa.int64()), (col(base // g1), pa.int64())]
    if kind == "wide2":
        g1 = max(1, int(G ** 0.5))
        return [(col((base % g1) * (1 << 44) - (1 << 61)), pa.int64()), (col((base // g1) * (1 << 40) + 12345), pa.int64())]
    if kind == "wide3hi":   # three wide columns, each with as many distinct values as there are groups: > 63 bits even as dictionary codes
        return [(col(base * 977_000_003 - (1 << 61)), pa.int64()), (col((base ^ 0x5DEECE66D) * 1_000_003 + 17), pa.int64()),
                (col(base * (1 << 33) - 99), pa.int64())]
    if kind == "i64n":
        return [(col(base * 977 - 5, True), pa.int64())]
    if kind == "i64x2n":
        g1 = max(1, int(G ** 0.5))
        return [(col(base % g1, True), pa.int64()), (col(base // g1), pa.int64())]
    raise ValueError(kind)


PROGRAMS = {
    "count*": lambda: ([(L.COUNT_STAR, None, None)], [None]),
    "sum_f64": lambda: ([(L.SUM, 1, pa.float64())], [col(f64)]),
    "sum_avg_f64": lambda: ([(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())], [col(f64)] * 2),
    "sum_i64": lambda: ([(L.SUM, 1, pa.int64())], [col(i64)]),
    "minmax_i64": lambda: ([(L.MIN, 1, pa.int64()), (L.MAX, 1, pa.int64())], [col(i64)] * 2),
    "avg_i32": lambda: ([(L.AVG, 1, pa.int32())], [col(i32)]),
    "sum_f32": lambda: ([(L.SUM, 1, pa.float32())], [col(f32)]),
    "sum_f64+max_i64": lambda: ([(L.SUM, 1, pa.float64()), (L.MAX, 2, pa.int64())], [col(f64), col(i64)]),
    "sum_f64+sum_i64+min_i32": lambda: ([(L.SUM, 1, pa.float64()), (L.SUM, 2, pa.int64()), (L.MIN, 3, pa.int32())], [col(f64), col(i64), col(i32)]),
    "7cols": lambda: ([(L.SUM, 1, pa.float64()), (L.SUM, 2, pa.int64()), (L.MIN, 3, pa.int32()), (L.AVG, 4, pa.float32()), (L.MAX, 5, pa.float64()),
                       (L.SUM, 6, pa.int64()), (L.AVG, 7, pa.float64()), (L.COUNT_STAR, None, None)],
                      [col(f64), col(i64), col(i32), col(f32), col(f64b), col(i64b), col(f64c), None]),
    "12cols": lambda: ([(L.SUM, 1 + j, pa.float64() if j % 2 == 0 else pa.int64()) for j in range(12)],
                       [col((f64, i64, f64b, i64b, f64c, i64c)[j % 6]) for j in range(12)]),
    "count_f64+count*": lambda: ([(L.COUNT, 1, pa.float64()), (L.COUNT_STAR, None, None)], [col(f64), None]),
}
if NULLS:
    PROGRAMS = {
        "count*": PROGRAMS["count*"],
        "sum_f64": PROGRAMS["sum_f64"],
        "sum_f64n": lambda: ([(L.SUM, 1, pa.float64())], [col(f64, True)]),
        "sum_avg_f64n": lambda: ([(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())], [col(f64, True)] * 2),
        "minmax_i64n": lambda: ([(L.MIN, 1, pa.int64()), (L.MAX, 1, pa.int64())], [col(i64, True)] * 2),
        "count_f64n+count*": lambda: ([(L.COUNT, 1, pa.float64()), (L.COUNT_STAR, None, None)], [col(f64, True), None]),
        "sum_f64n+max_i64": lambda: ([(L.SUM, 1, pa.float64()), (L.MAX, 2, pa.int64())], [col(f64, True), col(i64)]),
        "avg_i32n": lambda: ([(L.AVG, 1, pa.int32())], [col(i32, True)]),
    }
PREDS = {"none": None, "f64>": ("f64", ">", 32.0), "i32>": ("i32", ">", 0)}
pred_cols = {"f64": col(f64, NULLS), "i32": col(i32)}

rows = []
KINDS = ["i64", "i64n", "i64x2n"] if NULLS else ["i64", "i32", "f64", "i64x2", "wide2", "wide3hi"]
import os
if os.environ.get("CLIFF_KINDS"):
    KINDS = os.environ["CLIFF_KINDS"].split(",")
for kind, G in itertools.product(KINDS, [10, 10_000, 1_000_000, 20_000_000]):
    ks = keyset(kind, G)
    for pname, mk in PROGRAMS.items():
        for prname, pr in PREDS.items():
            if prname != "none" and pname not in ("sum_f64", "sum_avg_f64", "minmax_i64", "sum_f64+max_i64", "sum_f64n", "sum_avg_f64n", "minmax_i64n"):
                continue
            spec, inputs = mk()
            ms = None
            try:
                for rep in range(3):      # the first run sizes the allocator's blocks; the better of the next two counts
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL if len(ks) == 1 else L.MULTI_NUMERICAL, [t for _, t in ks], spec)
                    if pr:
                        agg.set_predicate(pr[1], pr[2])
                    agg.next([c for c, _ in ks], inputs, pred=pred_cols[pr[0]] if pr else None, nrows=n)
                    ng = agg.finish()
                    torch.cuda.synchronize(); t = (time.perf_counter() - t0) * 1e3
                    ms = t if rep < 2 else min(ms, t)
                    agg.close()
            except Exception as e:   # noqa
                print("ERR", kind, G, pname, prname, repr(e)[:100], flush=True)
                continue
            rows.append((ms * 1e9 / n, kind, G, pname, prname, ng))
    del ks
rows.sort(reverse=True)
for r in rows:
    print(f"{r[0]:9.1f} ms/1e9  key={r[1]:6s} G={r[2]:<9d} {r[3]:26s} pred={r[4]:5s} groups={r[5]}")
