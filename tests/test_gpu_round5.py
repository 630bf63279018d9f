"""Round 5 parity: reference-exact float MIN / MAX under NaNs and mixed-sign zeros (vnm_agg_exact.inc), stream memory bounds,
row-count guards."""
import os

import numpy as np
import pyarrow as pa
import pytest

from tests import util
from tests.test_gpu_agg import gpu_aggregate

pytestmark = pytest.mark.gpu
SEEDS = int(os.environ.get("VNM_FUZZ_SEEDS", "0")) or None


def _special_values(rng, n, dtype, p_nan, p_nz, p_pz):
    v = rng.normal(0.0, 50.0, n)
    v = np.round(v * 4) / 4          # ties at the extremes are common
    u = rng.random(n)
    v[u < p_nan] = np.nan
    v[(u >= p_nan) & (u < p_nan + p_nz)] = -0.0
    v[(u >= p_nan + p_nz) & (u < p_nan + p_nz + p_pz)] = 0.0
    if rng.random() < 0.3:           # zero IS the extreme of many groups
        v = np.where(np.isnan(v), v, np.abs(v) * (1.0 if rng.random() < 0.5 else -1.0))
        v[(u >= p_nan) & (u < p_nan + p_nz)] = -0.0
    return v.astype(dtype)


def _oracle(kind, groupby, funcs, batches, pred=None):
    from oracle import oracle as O
    o = O.OracleAggregate(kind, groupby, groupby, funcs)
    for b in batches:
        if pred is not None:
            b = O.filter_batch(b, O.cmp_mask(b.column(b.schema.names.index(pred[0])), O.GT, pred[2]))
        o.next(b)
    return o.result()


@pytest.mark.parametrize("seed", range(SEEDS or 48))
def test_ordered_min_max_fuzz_vs_oracle(seed):
    """MIN / MAX of float columns whose NaNs / -0.0 / +0.0 appear anywhere in the stream (also only in LATE batches: the
    operator then switches to the ordered mode in mid-stream and composes the prefix state with the suffix fold), NULL
    inputs, NULL keys, a fused WHERE, float32, one / two key columns / no GROUP BY, SUM / COUNT / integer MIN next to them.
    The oracle is the reference's row-at-a-time loop (bit-exact against oracle/_ref on minmax_ref.arrow)."""
    from oracle import oracle as O
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([300, 5000, 120_000, 700_000]))
    groups = int(rng.choice([1, 3, 40, 2000, 60_000]))
    kind = [O.SINGLE, O.SINGLE, O.MULTI, O.ONE_GROUP][seed % 4]
    dtype = np.float32 if seed % 5 == 3 else np.float64
    p_nan, p_nz, p_pz = [(0.02, 0.02, 0.05), (0.0, 0.03, 0.03), (0.05, 0.0, 0.0), (0.3, 0.1, 0.1), (0.001, 0.0, 0.2)][seed % 5]
    v = _special_values(rng, n, dtype, p_nan, p_nz, p_pz)
    w = _special_values(rng, n, np.float64, p_nan / 2, p_nz, p_pz)
    clean_prefix = int(rng.choice([0, 0, n // 3, n - 7 if n > 7 else 0]))   # specials only after this row: the switch falls in mid-stream
    if clean_prefix:
        keep = rng.normal(3.0, 50.0, clean_prefix)
        v[:clean_prefix] = np.where(keep == 0, 1.0, keep).astype(dtype)
        w[:clean_prefix] = np.abs(keep) + 1.0
    k1 = rng.integers(0, groups, n).astype(np.int64) * 13 - 7
    k2 = rng.integers(0, 3, n).astype(np.int32)
    i = rng.integers(-1000, 1000, n).astype(np.int64)
    cols = {"k1": pa.array(k1, mask=rng.random(n) < (0.01 if seed % 3 == 0 else 0.0)), "k2": pa.array(k2),
            "v": pa.array(v, mask=rng.random(n) < 0.03), "w": pa.array(w), "i": pa.array(i)}
    t = pa.table(cols)
    funcs = [(O.MIN, "v", "mn_v"), (O.MAX, "v", "mx_v"), (O.COUNT, "v", "c_v"), (O.MAX, "w", "mx_w"), (O.MIN, "w", "mn_w"),
             (O.MIN, "i", "mn_i"), (O.SUM, "i", "s_i"), (O.COUNT_STAR, "", "n")]
    if seed % 7 == 0:
        funcs = [(O.MAX, "v", "mx_v"), (O.MIN, "v", "mn_v")]
    groupby = {O.SINGLE: ["k1"], O.MULTI: ["k1", "k2"], O.ONE_GROUP: []}[kind]
    pred = ("w", ">", -20.0) if seed % 4 == 1 else None     # (NaN > x is False: the fused WHERE drops the NaN rows of w)
    chunk = int(rng.choice([n, max(1, n // 3), max(1, n // 11), 977]))
    batches = util.sliced_batches(t, chunk)
    got = gpu_aggregate(kind, groupby, groupby, funcs, batches, predicate=pred)
    exp = _oracle(kind, groupby, funcs, batches, pred)
    util.assert_batches_equal(got, exp, key_names=groupby, what=f"seed {seed}: n={n} G={groups} kind={kind} chunk={chunk} prefix={clean_prefix}")


def test_ordered_min_max_reference_vector_table_null():
    """The reference's own vector (vinum/tests/test_query_results.py:1270-1301, `test_table_null`): rows of one city are
    total = (NaN, 33.4, NaN): min(total) is NaN (the first row is), max(total) is NaN (the last row is)."""
    from oracle import oracle as O
    nan = float("nan")
    t = pa.table({"city": pa.array([1, 2, 1, 2, 1, 3], pa.int64()),
                  "total": pa.array([nan, 1.5, 33.4, nan, nan, None], pa.float64())})
    funcs = [(O.MIN, "total", "mn"), (O.MAX, "total", "mx"), (O.COUNT, "total", "c")]
    got = util.canon(gpu_aggregate(O.SINGLE, ["city"], ["city"], funcs, t.to_batches()), ["city"])
    mn, mx = got.column("mn").to_pylist(), got.column("mx").to_pylist()
    assert np.isnan(mn[0]) and np.isnan(mx[0])          # (NaN, 33.4, NaN)
    assert mn[1] == 1.5 and np.isnan(mx[1])             # (1.5, NaN): the NaN never beats 1.5 under MIN, replaces it under MAX
    assert mn[2] is None and mx[2] is None
    exp = _oracle(O.SINGLE, ["city"], funcs, t.to_batches())
    util.assert_batches_equal(got, exp, key_names=["city"], what="test_table_null")


def test_ordered_min_max_result_then_more_batches():
    """result() in mid-stream, more batches, result() again: the merged result of the ordered mode is rebuilt."""
    from oracle import oracle as O
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    rng = np.random.default_rng(11)
    n = 40_000
    k = rng.integers(0, 50, n).astype(np.int64)
    v = _special_values(rng, n, np.float64, 0.05, 0.05, 0.05)
    t = pa.table({"k": pa.array(k), "v": pa.array(v)})
    funcs = [(O.MIN, "v", "mn"), (O.MAX, "v", "mx"), (O.SUM, "v", "s")]
    batches = util.sliced_batches(t, 10_000)
    agg = ops.DeviceAggregate(O.SINGLE, [pa.int64()], [(f, 1, pa.float64()) for f, _, _ in funcs])
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for j, b in enumerate(batches):
        agg.next([DeviceColumn.from_arrow(b.column(0))], [DeviceColumn.from_arrow(b.column(1))] * 3, nrows=b.num_rows)
        o.next(b)
        if j in (1, 3):
            got = agg.result_arrays([0], ["k"], [f[2] for f in funcs])
            exp = o.result() if j == 3 else None
            if exp is not None:
                util.assert_col_equal(util.canon(got, ["k"]).column("mn"), util.canon(exp, ["k"]).column("mn"), "mn")
                util.assert_col_equal(util.canon(got, ["k"]).column("mx"), util.canon(exp, ["k"]).column("mx"), "mx")
    agg.close()
