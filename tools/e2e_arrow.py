"""End-to-end (PCIe-inclusive) rate of the Arrow-level boundary: host Arrow batches -> pinned staging -> HBM ->
fused aggregate -> Arrow result.  Reported in DESIGN.md next to the HBM-resident rate; never the bench value."""
import sys, time
import numpy as np, pyarrow as pa
sys.path.insert(0, ".")
from vinum_amd import vinum_lib as vl
n = 100_000_000
rng = np.random.default_rng(0)
t = pa.table({"k": rng.integers(0, 1000, n).astype(np.int64), "v": rng.integers(0, 2**14, n).astype(np.float64) / 128.0})
batches = t.to_batches(max_chunksize=1 << 24)
for rep in range(3):
    agg = vl.SingleNumericalHashAggregate(["k"], ["k"], [vl.AggFuncDef(vl.SUM, "v", "s"), vl.AggFuncDef(vl.AVG, "v", "a")])
    t0 = time.perf_counter()
    for b in batches:
        agg.next(b)
    res = agg.result()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n / dt / 1e6:.1f} Mrows/s, {16 * n / dt / 1e9:.2f} GB/s host->result, groups {res.num_rows}")
