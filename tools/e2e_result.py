"""result() of the Arrow-level boundary at a large group count: device finaliser (default) vs the host finaliser
(VNM_AGG_HOST_FINALIZE=1).  python tools/e2e_result.py"""
import sys, time
import numpy as np, pyarrow as pa
sys.path.insert(0, ".")
from vinum_amd import vinum_lib as vl
n, G = 60_000_000, 20_000_000
rng = np.random.default_rng(0)
t = pa.table({"k": rng.integers(0, G, n).astype(np.int64), "v": rng.integers(0, 2**14, n).astype(np.float64) / 128.0})
batches = t.to_batches(max_chunksize=1 << 24)
for rep in range(2):
    agg = vl.SingleNumericalHashAggregate(["k"], ["k"], [vl.AggFuncDef(vl.SUM, "v", "s"), vl.AggFuncDef(vl.AVG, "v", "a"), vl.AggFuncDef(vl.COUNT_STAR, "", "n")])
    for b in batches:
        agg.next(b)
    t0 = time.perf_counter()
    res = agg.result()
    print(f"rep {rep}: result() of {res.num_rows} groups x 4 columns: {(time.perf_counter() - t0) * 1e3:.0f} ms")
