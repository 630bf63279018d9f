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


def _check_null_table_vector(got: pa.Table):
    from tests.golden import planner_cases as P
    exp = P.NULL_TABLE_EXPECTED
    rows = {r["city_from"]: r for r in got.to_pylist()}
    for i, city in enumerate(exp["city_from"]):
        r = rows[city]
        for name, vals in exp.items():
            if name == "city_from" or name not in r:
                continue
            e, g = vals[i], r[name]
            if isinstance(e, float) and np.isnan(e):
                assert g is not None and np.isnan(g), f"{city} {name}: {g!r}, the reference's test expects NaN"
            else:
                assert g == e, f"{city} {name}: {g!r} != {e!r}"


def test_reference_null_vector_string_keys_through_vinum_lib():
    """vinum/tests/test_query_results.py:1270-1301 as the reference runs it: GROUP BY the STRING column city_from
    (GenericHashAggregate), count(*) / count(total) / count(name) / count(date) / count(is_vendor) / min / max / avg / sum
    (total) -- min(total) and max(total) of Berlin's (NaN, 33.4, NaN) are NaN.  Expected values: the reference's own vector."""
    from tests.golden import planner_cases as P
    from vinum_amd import vinum_lib as V
    t = P.null_table()
    funcs = [V.AggFuncDef(V.COUNT_STAR, "", "cnt_all"), V.AggFuncDef(V.COUNT, "total", "cnt_total"), V.AggFuncDef(V.COUNT, "name", "cnt_name"),
             V.AggFuncDef(V.COUNT, "date", "cnt_date_str"), V.AggFuncDef(V.COUNT, "is_vendor", "cnt_bool"),
             V.AggFuncDef(V.MIN, "total", "min_total"), V.AggFuncDef(V.MAX, "total", "max_total"),
             V.AggFuncDef(V.AVG, "total", "avg_total"), V.AggFuncDef(V.SUM, "total", "sum_total")]
    for chunk in (8, 3, 1):     # one batch; the NaNs of Berlin in different batches; a batch per row
        agg = V.GenericHashAggregate(["city_from"], ["city_from"], funcs)
        for b in util.sliced_batches(t, chunk):
            agg.next(b)
        _check_null_table_vector(pa.Table.from_batches([agg.result()]))


def test_reference_null_vector_string_keys_through_the_planner():
    from tests.golden import planner_cases as P
    from vinum_amd import planner
    fn = P.fn
    q = dict(select=["city_from", fn("count_star"), fn("count", "total"), fn("count", "name"), fn("min", "total"), fn("max", "total"),
                     fn("avg", "total"), fn("sum", "total")],
             aliases=[None, "cnt_all", "cnt_total", "cnt_name", "min_total", "max_total", "avg_total", "sum_total"],
             group_by=["city_from"])
    got = planner.execute(q, P.null_table())
    assert got.num_rows == 4
    _check_null_table_vector(got)


@pytest.mark.parametrize("seed", range(SEEDS or 40))
def test_random_plans_with_nans_and_signed_zeros_vs_oracle(seed, monkeypatch):
    """The dispatch-space fuzzer of test_random_plans_vs_oracle (1-3 key columns of mixed widths with NULLs -> packed /
    dictionary-coded / tuple keys, typed inputs, hints, predicates, skew, several batches, stream mode) with NaNs and signed
    zeros under the float MIN / MAX functions: whatever path the operator picks before and after it switches to the ordered
    mode, the result equals the reference's row-order dependent rule."""
    from oracle import oracle as O
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    rng = np.random.default_rng(91_000 + seed)
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "1000")
    cols, key_names, in_names, funcs, n, groups, skew = util.random_agg_case(rng, specials=True)
    pred = None
    if rng.random() < 0.3 and in_names:
        pred = (in_names[0], ">", 5 if pa.types.is_integer(cols[in_names[0]].type) else 5.0)
    t = pa.table(cols)
    names = t.schema.names
    kind = O.SINGLE if len(key_names) == 1 else O.MULTI
    hint = groups if rng.random() < 0.5 else 0
    batches = util.sliced_batches(t, int(rng.choice([n, n // 2 + 1, n // 5 + 1])))
    stream_mode = bool(rng.random() < 0.5)
    fspec = [(f, names.index(col) if col else None, t.schema.field(col).type if col else None) for f, col, _ in funcs]
    agg = ops.DeviceAggregate(kind, [t.schema.field(k).type for k in key_names], fspec, expected_groups=hint, stream_mode=stream_mode)
    if pred:
        agg.set_predicate(pred[1], pred[2])
    for b in batches:
        dc = {nm: DeviceColumn.from_arrow(b.column(j)) for j, nm in enumerate(names)}
        agg.next([dc[k] for k in key_names], [dc[col] if col else None for _, col, _ in funcs], pred=dc[pred[0]] if pred else None, nrows=b.num_rows)
    got = agg.result_arrays(list(range(len(key_names))), key_names, [f[2] for f in funcs])
    agg.close()
    exp = _oracle(kind, key_names, funcs, batches, pred)
    util.assert_agg_equal(got, exp, funcs, key_names, source=batches if pred is None else None,
                          what=f"seed {seed}: keys {[str(cols[k].type) for k in key_names]} inputs {[str(cols[v].type) for v in in_names]} "
                               f"G~{groups} skew={skew} hint={hint} pred={pred} stream={stream_mode} batches={len(batches)}")
