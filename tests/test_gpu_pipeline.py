"""The five BASELINE.json configs at small N through the build's own operator chain (vinum_amd.query.select),
checked against the oracle / pyarrow."""
import io

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.csv as pacsv
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _taxi_csv(n, seed=0):
    """config 1 input: taxi schema (README.rst:121-139, test_io.py:7-15), passenger_count in 0..6 with the
    empirical weights of vinum/tests/test_io.py:18."""
    rng = np.random.default_rng(seed)
    w = np.array([165, 34808, 7386, 2183, 1016, 3453, 989], float)
    pcnt = rng.choice(7, n, p=w / w.sum())
    fare = np.round(rng.lognormal(2.2, 0.6, n), 2)
    t = pa.table({"key": pa.array([f"k{i}" for i in range(n)]), "fare_amount": fare,
                  "pickup_longitude": rng.normal(-73.9, 0.1, n), "passenger_count": pcnt.astype(np.int64)})
    buf = io.BytesIO()
    pacsv.write_csv(t, buf)
    return buf.getvalue(), t


def test_config1_groupby_passenger_count_stream_csv():
    from vinum_amd.query import select
    from vinum_amd.core import AggregateFunction as F
    # BASELINE.json configs[0] at its own size: the 1M-row CSV, `SELECT passenger_count, count(*) ... GROUP BY passenger_count`
    # (+ avg(fare_amount)), against the ORACLE fed with the same record batches the CSV reader delivers: keys and counts bit for
    # bit, the float AVG held to the exact-mean bound (util.assert_agg_equal with `source`)
    from oracle import oracle as O
    data, t = _taxi_csv(1_000_000)
    reader = pacsv.open_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(block_size=1 << 20))
    got = select(reader, columns=["passenger_count"], group_by=["passenger_count"],
                 aggregates=[F("count", None, "cnt"), F("avg", "fare_amount", "avg_fare")])
    funcs = [(O.COUNT_STAR, "", "cnt"), (O.AVG, "fare_amount", "avg_fare")]
    o = O.OracleAggregate(O.SINGLE, ["passenger_count"], ["passenger_count"], funcs)
    fed = []
    for b in pacsv.open_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(block_size=1 << 20)):
        b = b.select(["fare_amount", "passenger_count"])
        o.next(b)
        fed.append(b)
    util.assert_agg_equal(got.combine_chunks().to_batches()[0], o.result(), funcs, ["passenger_count"], exact_float_inputs=(),
                          source=fed, what="config1")


def test_config2_filter():
    from vinum_amd.query import select
    rng = np.random.default_rng(2)
    n = 300_000
    t = pa.table({"fare_amount": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0, mask=rng.random(n) < 0.01),
                  "id": np.arange(n)})
    got = select(t, columns=["fare_amount", "id"], where=("fare_amount", ">", 64.0))
    m = pc.fill_null(pc.greater(t.column("fare_amount"), 64.0), False)
    util.assert_batches_equal(got.combine_chunks().to_batches()[0], t.filter(m).combine_chunks().to_batches()[0], what="config2")


def test_config3_filter_groupby():
    from oracle import oracle as O
    from vinum_amd.query import select
    from vinum_amd.core import AggregateFunction as F
    rng = np.random.default_rng(3)
    n = 400_000
    t = pa.table({"k": rng.integers(0, 50_000, n).astype(np.int64), "v": rng.integers(0, 2**14, n).astype(np.float64) / 128.0})
    got = select(t, columns=["k"], where=("v", ">", 64.0), group_by=["k"],
                 aggregates=[F("sum", "v", "s"), F("avg", "v", "a")], expected_groups=50_000)
    funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a")]
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in t.to_batches():
        o.next(O.filter_batch(b, O.cmp_mask(b.column(1), O.GT, 64.0)))
    util.assert_agg_equal(got.combine_chunks().to_batches()[0], o.result(), funcs, ["k"], what="config3")


def test_config5_orderby_limit_projection():
    from oracle import oracle as O
    from vinum_amd.query import select
    rng = np.random.default_rng(5)
    n = 500_000
    v = rng.normal(11, 9, n)
    v[rng.random(n) < 0.001] = np.nan
    t = pa.table({"v": pa.array(v, mask=rng.random(n) < 0.001), "a": rng.normal(0, 3, n), "b": rng.lognormal(0, 1, n)})
    got = select(t, columns=["v", ("e1", ("add", ("mul", "v", 2), 1)), ("e2", ("sub", "v", "a")), ("e3", ("mul", "a", "b"))],
                 order_by=["v"], sort_order=[1], limit=1000)
    s = O.OracleSort(["v"], [1])
    for b in t.to_batches():
        s.next(b)
    top = s.sorted().slice(0, 1000)
    nv = top.column(0).to_numpy(zero_copy_only=False)
    na, nb = top.column(1).to_numpy(), top.column(2).to_numpy()
    util.assert_col_equal(got.column("v").combine_chunks(), top.column(0), "v")
    for name, ref in [("e1", nv * 2 + 1), ("e2", nv - na), ("e3", na * nb)]:
        assert np.array_equal(got.column(name).to_numpy().view(np.uint64), ref.view(np.uint64)), name


def test_where_expression_trees_match_numpy_and_arrow():
    """General WHERE trees: comparisons between columns / literals, AND / OR / NOT,
    IS [NOT] NULL, BETWEEN, IN -- evaluated exactly as the reference's callables do it (NumPy on NaN-converted
    columns, vinum/core/expressions.py:27-48), then RecordBatch.filter."""
    from vinum_amd.query import select
    rng = np.random.default_rng(11)
    n = 200_000
    t = pa.table({
        "a": pa.array(rng.integers(-50, 50, n).astype(np.int64), mask=rng.random(n) < 0.05),
        "b": pa.array(rng.integers(-50, 50, n).astype(np.int64)),
        "x": pa.array(np.round(rng.normal(0, 10, n), 1), mask=rng.random(n) < 0.05),
        "id": pa.array(np.arange(n, dtype=np.int64)),
    })
    A = t.column("a").to_numpy(zero_copy_only=False)   # NULL -> NaN (record_batch.py:112-118)
    B = t.column("b").to_numpy()
    X = t.column("x").to_numpy(zero_copy_only=False)
    a_null = np.array(t.column("a").is_null())
    cases = [
        (("and", ("gt", "a", 5), ("lt", "x", 3.5)), (A > 5) & (X < 3.5)),
        (("or", ("ge", "a", "b"), ("is_null", "a")), (A >= B) | a_null),
        (("not", ("eq", "b", 7)), ~(B == 7)),
        (("between", "x", -2.5, 4), np.logical_and(X >= -2.5, X <= 4)),
        (("not_between", "a", -10, 10), np.logical_or(A < -10, A > 10)),
        (("in", "b", [1, 2, 3, 40]), np.isin(B, [1, 2, 3, 40])),
        (("not_in", "a", [0, 1]), np.isin(A, [0, 1], invert=True)),
        (("and", ("is_not_null", "x"), ("ne", ("add", "a", "b"), 0), ("gt", ("mul", "x", 2), "b")),
         np.array(t.column("x").is_valid()) & ((A + B) != 0) & ((X * 2) > B)),
    ]
    for expr, mask in cases:
        got = select(t, columns=["a", "b", "x", "id"], where=expr)
        exp = t.filter(pa.array(mask))
        util.assert_batches_equal(got.combine_chunks().to_batches()[0] if got.num_rows else got.to_batches()[0] if got.to_batches() else exp.slice(0, 0).combine_chunks().to_batches()[0],
                                  exp.combine_chunks().to_batches()[0], what=str(expr))


def test_pool_trim_releases_cached_blocks_only():
    """vnm_pool_trim: freed blocks are cached for reuse (a same-size request gets the same block back), trim hands them to the
    device and reports their bytes, and live blocks keep their contents."""
    from vinum_amd.device import DeviceBuffer, pool_trim
    pool_trim()
    live = DeviceBuffer.from_host(np.arange(1 << 16, dtype=np.int64))
    a = DeviceBuffer(48 << 20)
    ptr = a.ptr
    a.free()
    b = DeviceBuffer(48 << 20)
    assert b.ptr == ptr                      # reuse without going to the device
    b.free()
    released = pool_trim()
    assert released >= 48 << 20
    assert pool_trim() == 0                  # nothing cached any more
    np.testing.assert_array_equal(live.to_host(np.int64, 1 << 16), np.arange(1 << 16, dtype=np.int64))
    live.free()


def test_pool_gives_idle_cache_back_by_itself():
    """a host that never calls vnm_pool_trim must not sit on the cache -- the reaper thread releases cached
    blocks beyond keep_bytes once the allocator has been idle for idle_ms; an active allocator is left alone."""
    import time
    from vinum_amd import _lib as L
    from vinum_amd.device import DeviceBuffer, pool_trim
    lib = L.lib()
    pool_trim()
    try:
        lib.vnm_pool_set_idle_trim(400, 8 << 20)
        live = DeviceBuffer.from_host(np.arange(1 << 16, dtype=np.int64))
        for _ in range(3):
            DeviceBuffer(96 << 20).free()
            DeviceBuffer(40 << 20).free()
        assert lib.vnm_pool_cached_bytes() >= 136 << 20          # cached, the allocator is busy
        t0 = time.time()
        while time.time() - t0 < 0.3:                             # ... and stays so while it is being used
            DeviceBuffer(4096).free()
            time.sleep(0.01)
        assert lib.vnm_pool_cached_bytes() >= 136 << 20
        time.sleep(1.5)                                           # idle: the reaper trims down to keep_bytes
        assert lib.vnm_pool_cached_bytes() <= 8 << 20
        np.testing.assert_array_equal(live.to_host(np.int64, 1 << 16), np.arange(1 << 16, dtype=np.int64))
        live.free()
    finally:
        lib.vnm_pool_set_idle_trim(2000, 256 << 20)
