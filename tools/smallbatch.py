"""Arrow-level boundary fed with the reference's default batch size (10 000 rows, vinum/__init__.py:52)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, pyarrow as pa
from vinum_amd import vinum_lib as V
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
rng = np.random.default_rng(1)
t = pa.table({"k": rng.integers(0, 1000, n).astype(np.int64), "v": rng.integers(0, 2**14, n).astype(np.float64) / 128.0})
defs = [V.AggFuncDef(V.AggFuncType.SUM, "v", "s"), V.AggFuncDef(V.AggFuncType.AVG, "v", "a")]
for rep in range(2):
    rd = V.TableBatchReader(t); rd.set_batch_size(bs)
    op = V.SingleNumericalHashAggregate(["k"], ["k"], defs)
    t0 = time.perf_counter(); nb = 0
    while True:
        b = rd.next()
        if b is None: break
        op.next(b); nb += 1
    r = op.result()
    dt = time.perf_counter() - t0
    print(f"{n:.3g} rows in {nb} batches of {bs}: {dt*1e3:.0f} ms = {n/dt/1e6:.0f} Mrows/s, {r.num_rows} groups")
