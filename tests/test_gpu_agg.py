"""GPU parity: hash group-by aggregate (HIP, through the C ABI) vs golden vectors of the real reference,
the reference's gtest known answers, and the oracle on seeded inputs."""
import ctypes
import os

import numpy as np
import pyarrow as pa
import pytest

from tests import util
from tests.golden import gtest_fixtures as G

pytestmark = pytest.mark.gpu
MAN = util.manifest()


def gpu_aggregate(kind, groupby, agg_cols, funcs, batches, predicate=None, expected_groups=0) -> pa.RecordBatch:
    """Drive the device-level C ABI the way AggregateOperator drives the reference classes
    (vinum/core/aggregate.py:114-124): next(batch) per batch, one result()."""
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    schema = batches[0].schema
    key_types = [schema.field(c).type for c in groupby]
    names = schema.names
    fspec = [(f, names.index(col) if col else None, schema.field(col).type if col else None) for f, col, _ in funcs]
    agg = ops.DeviceAggregate(kind, key_types, fspec, expected_groups=expected_groups)
    if predicate:
        agg.set_predicate(predicate[1], predicate[2])
    for b in batches:
        cache = {}
        def dev(name):
            if name not in cache:
                cache[name] = DeviceColumn.from_arrow(b.column(names.index(name)))
            return cache[name]
        keys = [dev(c) for c in groupby]
        inputs = [dev(col) if col else None for _, col, _ in funcs]
        pred = dev(predicate[0]) if predicate else None
        agg.next(keys, inputs, pred=pred, nrows=b.num_rows)
    # The result finalised ON THE DEVICE first (vnm_agg_result_device_alloc): when the last batch took the dense-key path its
    # pending final pass writes the result columns itself; then the host finaliser over the dense partial state (which that
    # pass produces when asked again).  The two must agree bit for bit (rows in canonical order: the fused pass and the
    # partial-state pass hand out their output rows in their own orders).
    dcols = None
    try:
        dcols = agg.result_device([groupby.index(c) for c in agg_cols])
    except ops.NeedsHostFinalize:
        pass
    res = agg.result_arrays([groupby.index(c) for c in agg_cols], agg_cols, [f[2] for f in funcs])
    if dcols is not None:
        dev = pa.RecordBatch.from_arrays([c.to_arrow() for c in dcols], names=res.schema.names)
        util.assert_batches_equal(dev, res, key_names=list(agg_cols), what="device finalisation vs host finalisation")
    else:
        assert any(pa.types.is_decimal(f.type) for f in res.schema), "only a decimal128 promotion may need the host"
    agg.close()
    return res


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_gtest_known_answers(name):
    c = G.CASES[name]
    table = G.table_for(c)
    from oracle import oracle as O
    for kind in c["kinds"]:
        # the device level of the C ABI takes numeric columns; GenericHashAggregate and string aggregate inputs live one level up
        # (vinum_amd.vinum_lib: tests/test_gpu_vinum_lib.py runs these cases through it)
        if kind == G.GENERIC or any(col and not O._is_numeric_type(table.schema.field(col).type) for _, col, _ in c["funcs"]):
            continue
        table_n = table.select([f.name for f in table.schema if O._is_numeric_type(f.type)])
        res = gpu_aggregate(kind, c["groupby"], c["agg_cols"], c["funcs"], G.feed_batches(table_n))
        res = G.sort_result(res, c["sort_cols"])
        assert res.num_columns == len(c["expected"])
        for i, exp in enumerate(c["expected"]):
            # float SUM/AVG over <= 3 addends: order may differ from the reference -> 1 ULP (north_star tolerance)
            util.assert_col_equal(res.column(i), exp, f"{name}[{kind}] col {i}", ulps=1)


@pytest.mark.parametrize("case", MAN["agg"], ids=lambda c: c["name"])
def test_reference_golden(case):
    table = util.read_ipc(case["input"])
    expected = util.read_ipc(case["expected"])
    funcs = [tuple(f) for f in case["funcs"]]
    res = gpu_aggregate(case["kind"], case["groupby"], case["agg_cols"], funcs, util.sliced_batches(table, case["chunk"]))
    # (float SUM / AVG: the exact-sum bound needs the group keys in the result to line the groups up with their rows)
    util.assert_agg_equal(res, expected, funcs, case["agg_cols"], what=case["name"],
                          source=table if list(case["agg_cols"]) == list(case["groupby"]) else None)


@pytest.mark.parametrize("groups", [1, 7, 1000, 5000, 200_000])
@pytest.mark.parametrize("with_pred", [False, True])
def test_filter_groupby_vs_oracle(groups, with_pred):
    """config-3 shape: SELECT k, sum(v), avg(v), count(*) [WHERE v > X] GROUP BY k on quantised values
    (every partial sum exactly representable -> bit-exact float aggregates)."""
    from oracle import oracle as O
    rng = np.random.default_rng(groups)
    n = 600_000
    k = rng.integers(0, groups, n).astype(np.int64) * 7919 - 1000
    v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    t = pa.table({"k": pa.array(k, mask=rng.random(n) < 0.01), "v": pa.array(v, mask=rng.random(n) < 0.05)})
    funcs = [(O.SUM, "v", "sum_v"), (O.AVG, "v", "avg_v"), (O.COUNT_STAR, "", "n"), (O.MIN, "v", "min_v"),
             (O.MAX, "v", "max_v"), (O.COUNT, "v", "cnt_v")]
    batches = util.sliced_batches(t, 250_000)
    pred = ("v", ">", 64.0) if with_pred else None
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred)
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in batches:
        if with_pred:
            mask = O.cmp_mask(b.column(1), O.GT, 64.0)
            b = O.filter_batch(b, mask)
        o.next(b)
    util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"G={groups} pred={with_pred}")


def test_key_identity_rules():
    """+0.0 / -0.0 are different groups, NaNs group by payload, NULL is its own group (emitted last on the
    single-key path), the all-ones key pattern is a legal key."""
    from oracle import oracle as O
    nan1 = np.frombuffer(np.uint64(0x7FF8000000000001).tobytes(), np.float64)[0]
    nan2 = np.frombuffer(np.uint64(0x7FF8000000000002).tobytes(), np.float64)[0]
    allones = np.frombuffer(np.uint64(0xFFFFFFFFFFFFFFFF).tobytes(), np.float64)[0]
    kf = np.array([0.0, -0.0, nan1, nan2, nan1, 1.5, allones, allones, 0.0, -0.0] * 50)
    mask = np.zeros(len(kf), bool); mask[::7] = True
    v = np.arange(len(kf), dtype=np.int64)
    t = pa.table({"k": pa.array(kf, mask=mask), "v": pa.array(v)})
    funcs = [(O.COUNT_STAR, "", "n"), (O.SUM, "v", "s")]
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, t.to_batches())
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in t.to_batches():
        o.next(b)
    exp = o.result()
    util.assert_agg_equal(got, exp, funcs, ["k"], what="key identity")
    assert got.column(0)[got.num_rows - 1].as_py() is None  # NULL group last

    ki = np.array([-1, 0, 2**63 - 1, -2**63, -1, 5], dtype=np.int64)
    t2 = pa.table({"k": pa.array(ki), "v": pa.array(np.arange(6, dtype=np.int64))})
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, t2.to_batches())
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in t2.to_batches():
        o.next(b)
    util.assert_agg_equal(got, o.result(), funcs, ["k"], what="int key -1 == EMPTY sentinel")


def test_empty_inputs():
    from oracle import oracle as O
    empty = pa.RecordBatch.from_arrays([pa.array([], pa.int64()), pa.array([], pa.float64())], names=["k", "v"])
    funcs = [(O.COUNT_STAR, "", "n"), (O.SUM, "v", "s"), (O.MIN, "v", "mn")]
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, [empty])
    assert got.num_rows == 0
    got = gpu_aggregate(O.ONE_GROUP, [], [], funcs, [empty])
    assert got.to_pydict() == {"n": [0], "s": [None], "mn": [None]}


@pytest.mark.parametrize("groups", [5_000, 200_000])
@pytest.mark.parametrize("levels", [1, 2])
@pytest.mark.parametrize("with_pred", [False, True])
def test_partitioned_path_vs_oracle(groups, levels, with_pred, monkeypatch):
    """Large-G path: rows are radix partitioned (1 or 2 levels) and aggregated per partition in LDS.
    Same query shape as BASELINE configs[2]; quantised values -> bit-exact; two batches exercise the
    run -> table merge; the all-ones key (the table's EMPTY sentinel) must survive."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_PART_L1_MAX", "4" if levels == 2 else "256")
    rng = np.random.default_rng(groups + levels)
    n = 700_001
    k = rng.integers(0, groups, n).astype(np.int64) * 1_000_003 - 77
    k[rng.integers(0, n, 5)] = -1
    v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    t = pa.table({"k": pa.array(k), "v": pa.array(v)})
    funcs = [(O.SUM, "v", "sum_v"), (O.AVG, "v", "avg_v"), (O.COUNT_STAR, "", "n"), (O.COUNT, "v", "cnt_v")]
    pred = ("v", ">", 64.0) if with_pred else None
    for batches in (t.to_batches(), util.sliced_batches(t, 400_000)):
        got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred, expected_groups=groups)
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
        for b in batches:
            if with_pred:
                b = O.filter_batch(b, O.cmp_mask(b.column(1), O.GT, 64.0))
            o.next(b)
        util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"partitioned G={groups} L={levels} pred={with_pred}")


@pytest.mark.parametrize("shape", ["count_star", "minmax_f64", "int64_mix", "uint64_mix", "pred_on_other_column"])
@pytest.mark.parametrize("levels", [1, 2])
def test_partitioned_path_generic_programs(shape, levels, monkeypatch):
    """The partitioned path with a GENERIC accumulator program (anything over at most one 8-byte input column):
    entries carry (key, raw value bits), the final pass interprets them.  COUNT(*) alone (configs[0]'s query
    shape), MIN/MAX on the order-preserving encodings, int64 / uint64 SUM and AVG through the 128-bit lanes
    (values near the int64 range so the decimal128 promotion triggers).  Bit-exact against the oracle."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_PART_L1_MAX", "4" if levels == 2 else "256")
    rng = np.random.default_rng(len(shape) * 7 + levels)
    n = 500_003
    groups = 60_000
    k = rng.integers(0, groups, n).astype(np.int64) * 1_000_003 - 77
    k[rng.integers(0, n, 5)] = -1
    cols = {"k": pa.array(k)}
    pred = None
    if shape == "count_star":
        funcs = [(O.COUNT_STAR, "", "n")]
    elif shape == "minmax_f64":
        v = rng.normal(size=n) * 1e3
        v[rng.integers(0, n, 50)] = np.nan
        v[rng.integers(0, n, 50)] = -0.0
        cols["v"] = pa.array(v)
        funcs = [(O.MIN, "v", "mn"), (O.MAX, "v", "mx"), (O.COUNT, "v", "c")]
        pred = ("v", "<", 500.0)
    elif shape == "int64_mix":
        v = rng.integers(-2**62, 2**62, n).astype(np.int64)
        cols["v"] = pa.array(v)
        funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.MIN, "v", "mn"), (O.MAX, "v", "mx"), (O.COUNT_STAR, "", "n")]
    elif shape == "uint64_mix":
        v = rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2)
        cols["v"] = pa.array(v)
        funcs = [(O.SUM, "v", "s"), (O.MAX, "v", "mx"), (O.MIN, "v", "mn")]
    else:
        cols["v"] = pa.array(rng.integers(-1000, 1000, n).astype(np.int64))
        cols["p"] = pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)
        funcs = [(O.SUM, "v", "s"), (O.COUNT_STAR, "", "n")]
        pred = ("p", ">", 64.0)
    t = pa.table(cols)
    names = t.schema.names
    for batches in (t.to_batches(), util.sliced_batches(t, 300_000)):
        got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred, expected_groups=groups)
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
        for b in batches:
            if pred:
                op = {"<": O.LT, ">": O.GT}[pred[1]]
                b = O.filter_batch(b, O.cmp_mask(b.column(names.index(pred[0])), op, pred[2]))
            o.next(b)
        util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"partitioned generic {shape} L={levels}")


def test_partitioned_path_wrong_hint_falls_back():
    """A hint far below the real group count overflows the per-partition LDS tables; the operator must
    notice and produce the right answer through the general path."""
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    n = 900_000
    t = pa.table({"k": pa.array(rng.integers(0, 800_000, n).astype(np.int64)),
                  "v": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)})
    funcs = [(O.SUM, "v", "sum_v"), (O.COUNT_STAR, "", "n")]
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, t.to_batches(), expected_groups=3000)
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in t.to_batches():
        o.next(b)
    util.assert_agg_equal(got, o.result(), funcs, ["k"], what="wrong hint")


@pytest.mark.parametrize("ncols", [2, 3])
@pytest.mark.parametrize("levels", [1, 2])
@pytest.mark.parametrize("with_pred", ["none", "on_input", "on_other"])
def test_partitioned_path_several_input_columns(ncols, levels, with_pred, monkeypatch):
    """Aggregates over two or three 8-byte input columns take the partitioned path with wide entries
    (key + one raw value per column); every accumulator kind, a predicate on one of the inputs or on a
    fourth column.  Bit-exact against the oracle (quantised floats, integers)."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_PART_L1_MAX", "4" if levels == 2 else "256")
    rng = np.random.default_rng(ncols * 10 + levels)
    n = 450_007
    groups = 50_000
    k = rng.integers(0, groups, n).astype(np.int64) * 977 - 5
    k[rng.integers(0, n, 4)] = -1
    cols = {"k": pa.array(k),
            "a": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0),
            "b": pa.array(rng.integers(-2**40, 2**40, n).astype(np.int64)),
            "c": pa.array(rng.integers(0, 2**63, n).astype(np.uint64)),
            "p": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)}
    funcs = [(O.SUM, "a", "sa"), (O.AVG, "a", "ma"), (O.MAX, "b", "xb"), (O.SUM, "b", "sb"), (O.COUNT_STAR, "", "n")]
    if ncols == 3:
        funcs += [(O.MIN, "c", "nc"), (O.AVG, "c", "mc")]
    pred = {"none": None, "on_input": ("a", ">", 64.0), "on_other": ("p", "<=", 100.0)}[with_pred]
    t = pa.table(cols)
    names = t.schema.names
    for batches in (t.to_batches(), util.sliced_batches(t, 250_000)):
        got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred, expected_groups=groups)
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
        for b in batches:
            if pred:
                op = {">": O.GT, "<=": O.LE}[pred[1]]
                b = O.filter_batch(b, O.cmp_mask(b.column(names.index(pred[0])), op, pred[2]))
            o.next(b)
        util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"wide entries ncols={ncols} L={levels} pred={with_pred}")


@pytest.mark.parametrize("case", ["nullable_f64", "int32_and_nullable_i64", "float32_uint16", "int_predicate"])
@pytest.mark.parametrize("levels", [1, 2])
def test_partitioned_path_nulls_and_narrow_types(case, levels, monkeypatch):
    """Wide entries also carry a validity word and raw bits of any numeric width, and the first pass evaluates any
    predicate column: NULL inputs (skipped, all-NULL groups give NULL), int32 / float32 / uint16 inputs (AVG of
    16-bit ints is float32, agg_func_factory.cpp:179-196), an integer predicate."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_PART_L1_MAX", "4" if levels == 2 else "256")
    rng = np.random.default_rng(len(case) + levels)
    n = 400_003
    groups = 40_000
    k = rng.integers(0, groups, n).astype(np.int64) * 31 - 9
    cols = {"k": pa.array(k)}
    pred = None
    if case == "nullable_f64":
        cols["a"] = pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0, mask=rng.random(n) < 0.3)
        funcs = [(O.SUM, "a", "s"), (O.AVG, "a", "m"), (O.COUNT, "a", "c"), (O.COUNT_STAR, "", "n"), (O.MIN, "a", "lo")]
        pred = ("a", ">", 10.0)   # NULL compares False (NaN), record_batch.py:112-118
    elif case == "int32_and_nullable_i64":
        cols["a"] = pa.array(rng.integers(-2**31, 2**31 - 1, n).astype(np.int32))
        cols["b"] = pa.array(rng.integers(-2**50, 2**50, n).astype(np.int64), mask=rng.random(n) < 0.5)
        funcs = [(O.SUM, "a", "sa"), (O.AVG, "a", "ma"), (O.MAX, "a", "xa"), (O.SUM, "b", "sb"), (O.MIN, "b", "nb"), (O.COUNT, "b", "cb")]
    elif case == "float32_uint16":
        cols["a"] = pa.array((rng.integers(0, 2**10, n) / 8.0).astype(np.float32))
        cols["b"] = pa.array(rng.integers(0, 2**16, n).astype(np.uint16))
        funcs = [(O.SUM, "a", "sa"), (O.MAX, "a", "xa"), (O.AVG, "b", "mb"), (O.SUM, "b", "sb"), (O.MIN, "b", "nb")]
    else:
        cols["a"] = pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)
        cols["p"] = pa.array(rng.integers(-100, 100, n).astype(np.int64))
        funcs = [(O.SUM, "a", "s"), (O.COUNT_STAR, "", "n")]
        pred = ("p", ">=", 0)
    t = pa.table(cols)
    names = t.schema.names
    for batches in (t.to_batches(), util.sliced_batches(t, 250_000)):
        got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred, expected_groups=groups)
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
        for b in batches:
            if pred:
                op = {">": O.GT, ">=": O.GE}[pred[1]]
                b = O.filter_batch(b, O.cmp_mask(b.column(names.index(pred[0])), op, pred[2]))
            o.next(b)
        util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"wide entries {case} L={levels}")


@pytest.mark.parametrize("shape", ["sum_avg", "min_max_int", "count_only"])
@pytest.mark.parametrize("spill", [True, False])
def test_partitioned_path_skewed_keys(shape, spill, monkeypatch):
    """Power-law keys: a few keys own most rows, so their partition regions fill up in the first scatter pass.
    Entries that do not fit spill to the scan kernel (whose LDS table absorbs exactly the heavy keys); the result is
    a run + a table, merged at finish.  With spilling disabled the operator must notice the overflow right after the
    pass (not after aggregating doomed partitions) and fall back.  Two batches also exercise run + table + run."""
    import time
    from oracle import oracle as O
    if not spill:
        monkeypatch.setenv("VNM_AGG_NO_SPILL", "1")
    rng = np.random.default_rng(11)
    n = 900_000
    groups = 80_000
    k = np.floor(groups * rng.random(n) ** 8).astype(np.int64)
    k[rng.integers(0, n, 3)] = -1   # the EMPTY-sentinel key among them
    if shape == "sum_avg":
        t = pa.table({"k": pa.array(k), "v": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)})
        funcs = [(O.SUM, "v", "sum_v"), (O.AVG, "v", "avg_v"), (O.COUNT_STAR, "", "n")]
    elif shape == "count_only":   # key-only (8-byte) entries overflow a region, the operator goes back to 16-byte entries + spill
        t = pa.table({"k": pa.array(k)})
        funcs = [(O.COUNT_STAR, "", "n")]
    else:
        t = pa.table({"k": pa.array(k), "v": pa.array(rng.integers(-2**40, 2**40, n).astype(np.int64))})
        funcs = [(O.MIN, "v", "lo"), (O.MAX, "v", "hi"), (O.SUM, "v", "s"), (O.COUNT, "v", "c")]
    for batches in (t.to_batches(), util.sliced_batches(t, 500_000)):
        t0 = time.perf_counter()
        got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, expected_groups=groups)
        assert time.perf_counter() - t0 < 20.0
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
        for b in batches:
            o.next(b)
        util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"skewed keys {shape} spill={spill}")


@pytest.mark.parametrize("groups", [50, 20_000, 400_000])
def test_hintless_operator_estimates_cardinality(groups, monkeypatch):
    """No expected_groups: the operator samples the first batch (scratch table, then HyperLogLog), picks the LDS
    or the partitioned path itself, and still matches the oracle bit for bit."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "1000")
    rng = np.random.default_rng(groups)
    n = 800_000
    t = pa.table({"k": pa.array(rng.integers(0, groups, n).astype(np.int64) * 31 + 5),
                  "v": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)})
    funcs = [(O.SUM, "v", "sum_v"), (O.AVG, "v", "avg_v"), (O.COUNT_STAR, "", "n")]
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, t.to_batches(), predicate=("v", "<=", 100.0))
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in t.to_batches():
        o.next(O.filter_batch(b, O.cmp_mask(b.column(1), O.LE, 100.0)))
    util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"hintless G={groups}")


@pytest.mark.parametrize("world", [1, 2, 8])
def test_bucket_by_owner_and_merge_rows_roundtrip(world):
    """Multi-GPU building blocks on one GPU: the device bucketing must agree with distributed.owner_of, and
    merging every bucket back (what the owners do after the all_to_all) must reproduce the aggregate."""
    import torch
    from oracle import oracle as O
    from vinum_amd import distributed as D
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    rng = np.random.default_rng(world)
    n = 300_000
    k = rng.integers(-20_000, 20_000, n).astype(np.int64)
    v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    kt, vt = torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda()
    spec = [(O.SUM, 1, pa.float64()), (O.COUNT_STAR, None, None)]
    agg = ops.DeviceAggregate(O.SINGLE, [pa.int64()], spec)
    agg.next([DeviceColumn.from_torch(kt)], [DeviceColumn.from_torch(vt), None], nrows=n)
    ng = agg.finish()
    kw, aw = agg.layout()
    rows = torch.empty((ng, kw + aw), dtype=torch.int64, device="cuda")
    counts = agg.bucket_by_owner(world, rows.data_ptr())
    assert sum(counts) == ng
    own = D.owner_of([rows[:, j] for j in range(kw)], world).cpu()
    assert torch.equal(own, torch.repeat_interleave(torch.arange(world), torch.tensor(counts)))
    total = {}
    start = 0
    for o in range(world):                       # each owner merges its bucket into a fresh operator
        part = rows[start:start + counts[o]].contiguous()
        start += counts[o]
        m = ops.DeviceAggregate(O.SINGLE, [pa.int64()], spec)
        m.merge_rows(part.shape[0], part.data_ptr())
        res = m.result_arrays([0], ["k"], ["s", "n"])
        for kk, ss, nn in zip(res.column(0).to_pylist(), res.column(1).to_pylist(), res.column(2).to_pylist()):
            assert kk not in total
            total[kk] = (ss, nn)
    uk, inv = np.unique(k, return_inverse=True)
    exp_s = np.bincount(inv, weights=v)
    exp_n = np.bincount(inv)
    assert len(total) == len(uk)
    for i, kk in enumerate(uk.tolist()):
        assert total[kk] == (exp_s[i], exp_n[i])


@pytest.mark.parametrize("world", [1, 2, 8])
def test_partition_aligned_exchange_simulated(world):
    """Partition-aligned multi-GPU merge on one GPU: `world` operators play the ranks (same hint -> same hash
    partitions), their runs are reordered by partition, the blocks are routed by hand exactly as the all_to_all
    would, and every owner merges its partitions in LDS.  The union must equal the oracle on the union of the data."""
    import torch
    from oracle import oracle as O
    from vinum_amd import distributed as D
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    G = 300_000
    n = 400_000
    spec = [(O.SUM, 1, pa.float64()), (O.AVG, 1, pa.float64())]
    funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a")]
    aggs, tables, sends, pcs, counts = [], [], [], [], []
    for r in range(world):
        rng = np.random.default_rng(50 + r)
        k = rng.integers(0, G, n).astype(np.int64) * 13 - 7
        k[:3] = -1
        v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
        tables.append(pa.table({"k": k, "v": v}))
        kt, vt = torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda()
        # a generous hint gives >= 1024 final partitions, i.e. one workgroup per partition and a directory
        a = ops.DeviceAggregate(O.SINGLE, [pa.int64()], spec, expected_groups=2_000_000)
        a.next([DeviceColumn.from_torch(kt)], [DeviceColumn.from_torch(vt)] * 2, nrows=n)
        nfin = a.run_partitions()
        assert nfin > 0, "the partitioned path must have produced a partition-structured run"
        ng = a.finish()
        kw, aw = a.layout()
        send = torch.empty((ng, kw + aw), dtype=torch.int64, device="cuda")
        pc = torch.empty(nfin, dtype=torch.int32, device="cuda")
        counts.append(a.run_reorder(world, send.data_ptr(), pc.data_ptr()))
        assert sum(counts[-1]) == ng and int(pc.sum()) == ng
        aggs.append(a); sends.append(send); pcs.append(pc)
    bounds = [D.first_partition(o, nfin, world) for o in range(world + 1)]
    got_batches = []
    for o in range(world):                                  # what owner o receives
        blocks, pcr, offs = [], [], [0]
        for r in range(world):
            start = sum(counts[r][:o])
            blocks.append(sends[r][start:start + counts[r][o]])
            pcr.append(pcs[r][bounds[o]:bounds[o + 1]])
            offs.append(offs[-1] + counts[r][o])
        recv = torch.cat(blocks).contiguous()
        pc_recv = torch.cat(pcr).contiguous()
        m = ops.DeviceAggregate(O.SINGLE, [pa.int64()], spec)
        m.merge_partitioned(world, bounds[o + 1] - bounds[o], recv.data_ptr(), offs, pc_recv.data_ptr())
        got_batches.append(m.result_arrays([0], ["k"], ["s", "a"]))
    got = pa.Table.from_batches(got_batches).combine_chunks().to_batches()[0]
    o_ = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for t in tables:
        for b in t.to_batches():
            o_.next(b)
    util.assert_agg_equal(got, o_.result(), funcs, ["k"], what=f"partition-aligned world={world}")


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("values", ["quantised", "lognormal"])
def test_dense_table_exchange_simulated(world, values, monkeypatch):
    """Dense-table multi-GPU merge on one GPU (distributed.exchange_dense_tables routed by hand): `world` operators play the
    ranks; they agree on ONE key range (MIN / MAX of their sampled ranges), so their dense-path final passes write
    slot-compatible direct-addressed tables; owner o takes the slot range table_bounds()[o : o + 2] of every table and adds the
    slices up slot by slot.  The union of the owners' shards must equal the oracle on the union of the data -- bit for bit on
    quantised values, and the exactly rounded sum (math.fsum) on lognormal ones, whatever the number of ranks."""
    import math
    import torch
    from oracle import oracle as O
    from vinum_amd import distributed as D
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")   # the large-batch paths with test-sized inputs
    monkeypatch.setenv("VNM_DENSE_ONE_LEVEL", "0")              # ... and the two-level layout of the 1e8-group case at 2^21 codes
    G = 1_500_000
    n = 600_000
    spec = [(O.SUM, 1, pa.float64()), (O.AVG, 1, pa.float64()), (O.COUNT_STAR, None, None)]
    funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT_STAR, "", "n")]
    aggs, tables, cols = [], [], []
    ranges = []
    for r in range(world):
        rng = np.random.default_rng(70 + r)
        k = rng.integers(0, G, n).astype(np.int64) - 1_000_000 + 17 * r
        v = (rng.integers(0, 2**14, n).astype(np.float64) / 128.0) if values == "quantised" else rng.lognormal(3.0, 2.0, n)
        tables.append(pa.table({"k": k, "v": v}))
        kc, vc = DeviceColumn.from_torch(torch.from_numpy(k).cuda()), DeviceColumn.from_torch(torch.from_numpy(v).cuda())
        a = ops.DeviceAggregate(O.SINGLE, [pa.int64()], spec, expected_groups=G, rank_aligned=True)
        a.set_predicate(">", 1.0)
        ranges.append(a.dense_range(kc, n))
        aggs.append(a); cols.append((kc, vc))
    lo, hi = min(r[0] for r in ranges), max(r[1] for r in ranges)
    assert lo <= hi
    got_tabs = []
    for a, (kc, vc) in zip(aggs, cols):
        a.set_dense_range(lo, hi)
        a.next([kc], [vc, vc, None], pred=vc, nrows=n)
        got = a.dense_table()
        assert got is not None, "the batch must have gone through the dense path with its final pass pending"
        got_tabs.append(got)
    assert all(g[1:] == got_tabs[0][1:] for g in got_tabs), "every rank must derive the same code map"
    bits = got_tabs[0][1]
    bounds = D.table_bounds(1 << bits, world)
    views = [torch.as_tensor(D._RawView(g[0], 2 << bits), device="cuda").view(-1, 2) for g in got_tabs]
    got_batches = []
    for o in range(world):
        nloc = bounds[o + 1] - bounds[o]
        recv = torch.cat([t[bounds[o]:bounds[o + 1]] for t in views]).contiguous()
        m = ops.DeviceAggregate(O.SINGLE, [pa.int64()], spec)
        m.merge_dense_tables(aggs[0], [recv.data_ptr() + r * nloc * 16 for r in range(world)], bounds[o], nloc)
        got_batches.append(m.result_arrays([0], ["k"], ["s", "a", "n"]))
    got = pa.Table.from_batches(got_batches).combine_chunks().to_batches()[0]
    o_ = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for t in tables:
        for b in t.to_batches():
            o_.next(O.filter_batch(b, O.cmp_mask(b.column(1), O.GT, 1.0)))
    exp = o_.result()
    if values == "quantised":
        util.assert_agg_equal(got, exp, funcs, ["k"], what=f"dense tables world={world}")
    else:
        g, e = util.canon(got, ["k"]), util.canon(exp, ["k"])
        assert g.column(0).equals(e.column(0)) and g.column(2 + 1).equals(e.column(3))
        allk = np.concatenate([t.column("k").to_numpy() for t in tables])
        allv = np.concatenate([t.column("v").to_numpy() for t in tables])
        keep = allv > 1.0
        order = np.argsort(allk[keep], kind="stable")
        ks, vs = allk[keep][order], allv[keep][order]
        starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
        exact = np.array([math.fsum(vs[a:b].tolist()) for a, b in zip(starts, np.r_[starts[1:], len(vs)])])
        by_key = np.argsort(g.column(0).to_numpy(), kind="stable")     # (canon orders by bit pattern: negative keys last)
        assert np.array_equal(g.column(0).to_numpy()[by_key], ks[starts])
        assert np.array_equal(g.column(1).to_numpy()[by_key], exact), "float SUM must be the exactly rounded sum for every number of ranks"
    # the same operators still give their own partial state afterwards (the pending entries stay until the handle goes)
    assert aggs[0].finish() > 0


@pytest.mark.parametrize("scenario", ["dims", "nulls_and_negatives", "float_key", "demote_on_later_batch", "unpackable"])
def test_multi_key_packed_composite_keys(scenario):
    """Multi-column GROUP BY: key columns whose observed ranges fit 63 bits are packed into one word per row and run
    through the single-key machinery, result keys unpacked at the end; a later batch outside the ranges demotes the
    operator to the wide-key table; ranges that do not fit never pack.  All bit-exact against the oracle, NULL keys
    (null == null, multi_numerical_hash_aggregate.h:11-18) included."""
    from oracle import oracle as O
    rng = np.random.default_rng(len(scenario))
    n = 400_000
    v = pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)
    if scenario == "dims":
        a = pa.array(rng.integers(0, 1000, n).astype(np.int64))
        b = pa.array(rng.integers(0, 12, n).astype(np.int32))
        c = pa.array(rng.integers(0, 3, n).astype(np.uint8))
        keys = {"a": a, "b": b, "c": c}
    elif scenario == "nulls_and_negatives":
        a = pa.array(rng.integers(-500, 500, n).astype(np.int64), mask=rng.random(n) < 0.05)
        b = pa.array(rng.integers(-3, 4, n).astype(np.int16), mask=rng.random(n) < 0.2)
        keys = {"a": a, "b": b}
    elif scenario == "float_key":
        a = pa.array(rng.integers(0, 50, n).astype(np.float64) * 0.25 - 3.0)   # 50 distinct bit patterns, wide span
        b = pa.array(rng.integers(0, 40, n).astype(np.int64))
        keys = {"a": a, "b": b}
    elif scenario == "demote_on_later_batch":
        a = np.concatenate([rng.integers(0, 100, n // 2), rng.integers(10**12, 10**12 + 100, n - n // 2)]).astype(np.int64)
        b = pa.array(rng.integers(0, 7, n).astype(np.int64))
        keys = {"a": pa.array(a), "b": b}
    else:
        a = pa.array(rng.integers(-2**62, 2**62, n).astype(np.int64))
        b = pa.array(rng.integers(-2**62, 2**62, n).astype(np.int64))
        keys = {"a": a, "b": b}
    t = pa.table({**keys, "v": v})
    names = list(keys)
    funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "m"), (O.COUNT_STAR, "", "n"), (O.MIN, "v", "lo")]
    batches = util.sliced_batches(t, n // 2)
    for pred in (None, ("v", ">", 64.0)):
        got = gpu_aggregate(O.MULTI, names, names, funcs, batches, predicate=pred)
        o = O.OracleAggregate(O.MULTI, names, names, funcs)
        for bt in batches:
            if pred:
                bt = O.filter_batch(bt, O.cmp_mask(bt.column(len(names)), O.GT, 64.0))
            o.next(bt)
        util.assert_agg_equal(got, o.result(), funcs, names, what=f"packed multi-key {scenario} pred={pred}")


@pytest.mark.parametrize("program", ["count", "two_cols", "nullable_col", "three_cols_i32"])
@pytest.mark.parametrize("heavy", ["null_key_half", "one_value_half", "three_values_2pct"])
def test_heavy_keys_spill_from_wide_entries(program, heavy):
    """A key holding half of the rows (NULL keys, a default value) or a few keys with 2 % each, next to 60 000 ordinary groups:
    their partition regions fill up and the wide scatter kernels spill the entries (as the hot shape always did); the spilled
    entries come back as columns (spill_unzip_kernel) for the general scan, the rest stays on the partitioned path.  Before: one
    full region sent the whole batch to the per-row HBM-atomics path.  Bit-exact against the oracle."""
    import ctypes
    from oracle import oracle as O
    from vinum_amd import _lib as L
    rng = np.random.default_rng(len(program) * 31 + len(heavy))
    n, groups = 1_200_000, 60_000
    k = rng.integers(0, groups, n).astype(np.int64) * 977 - 12345
    kmask = None
    if heavy == "null_key_half":
        kmask = rng.random(n) < 0.5
    elif heavy == "one_value_half":
        k[rng.random(n) < 0.5] = 7
    else:
        r = rng.random(n)
        for j, val in enumerate((-1, 5, 2**40)):          # -1: the bit pattern of the tables' EMPTY word
            k[(r >= 0.02 * j) & (r < 0.02 * (j + 1))] = val
    cols = {"k": pa.array(k, mask=kmask)}
    cols["a"] = pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)
    cols["b"] = pa.array(rng.integers(-2**40, 2**40, n).astype(np.int64), mask=(rng.random(n) < 0.2) if program == "nullable_col" else None)
    cols["c"] = pa.array(rng.integers(-2**20, 2**20, n).astype(np.int32))
    funcs = {"count": [(O.COUNT_STAR, "", "n")],
             "two_cols": [(O.SUM, "a", "s"), (O.MAX, "b", "hi"), (O.COUNT_STAR, "", "n")],
             "nullable_col": [(O.SUM, "a", "s"), (O.MIN, "b", "lo"), (O.COUNT, "b", "nb")],
             "three_cols_i32": [(O.AVG, "a", "m"), (O.SUM, "b", "sb"), (O.MIN, "c", "lo"), (O.COUNT_STAR, "", "n")]}[program]
    t = pa.table(cols)
    batches = util.sliced_batches(t, n // 2)

    def launches(name):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        L.lib().vnm_profile_query(name, ctypes.byref(ms), ctypes.byref(cnt))
        return cnt.value
    for pred in (None, ("a", ">", 64.0)):
        L.lib().vnm_set_profiling(1)
        got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred, expected_groups=groups)
        fin = launches(b"agg_part_final")
        L.lib().vnm_set_profiling(0)
        assert fin >= len(batches), fin              # every batch finished on the partitioned path
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
        for bt in batches:
            if pred:
                bt = O.filter_batch(bt, O.cmp_mask(bt.column(1), O.GT, 64.0))
            o.next(bt)
        util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"heavy keys {heavy} {program} pred={pred}")


@pytest.mark.parametrize("ncols,nulls", [(3, False), (4, False), (5, False), (4, True), (5, True), (6, False), (6, True), (7, False),
                                         (8, True), (12, False), (14, True)])
def test_many_input_columns_take_the_partitioned_path(ncols, nulls):
    """SELECT k, sum(c1), min(c2), avg(c3), ... GROUP BY k with many groups: the partition entries carry the key and up to six
    input values (or five and a validity word), and the final pass shrinks its LDS table until the accumulator words fit
    (three float SUMs + counts used to exceed the LDS and fall back to per-row HBM atomics: 300 ms per 1e9 rows).  More columns
    than that (seven, or six with NULLs) split the program into sub-operators over the same key whose results are joined by
    key (round 3; before: the LDS scan and its flush storms).  Mixed column types, NULLs, a predicate, two batches; bit-exact
    against the oracle."""
    import ctypes
    from oracle import oracle as O
    from vinum_amd import _lib as L
    rng = np.random.default_rng(ncols * 2 + nulls)
    n, groups = 1_500_000, 60_000
    k = rng.integers(0, groups, n).astype(np.int64) * 977 - 12345
    cols = {"k": pa.array(k)}
    funcs = []
    kinds = [O.SUM, O.MIN, O.AVG, O.MAX, O.COUNT, O.SUM, O.MIN]
    for c in range(ncols):
        if c % 3 == 0:
            vals = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
        elif c % 3 == 1:
            vals = rng.integers(-2**31, 2**31, n).astype(np.int32)
        else:
            vals = rng.integers(-2**40, 2**40, n).astype(np.int64)
        mask = (rng.random(n) < 0.1) if (nulls and c % 2 == 0) else None
        cols[f"c{c}"] = pa.array(vals, mask=mask)
        funcs.append((kinds[c % len(kinds)], f"c{c}", f"f{c}"))
        if c % 4 == 1:
            funcs.append((O.AVG, f"c{c}", f"g{c}"))      # a second function of the same column: the two stay in one part
    funcs.insert(2, (O.COUNT_STAR, "", "n"))
    t = pa.table(cols)
    batches = util.sliced_batches(t, n // 2)
    split = ncols + (1 if nulls else 0) > 6

    def launches(name):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        L.lib().vnm_profile_query(name, ctypes.byref(ms), ctypes.byref(cnt))
        return cnt.value
    for pred in (None, ("c0", ">", 64.0)):
        L.lib().vnm_set_profiling(1)
        got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred, expected_groups=groups)
        p1, joins, scans = launches(b"agg_part_scatter1"), launches(b"agg_split_join"), launches(b"agg_scan")
        L.lib().vnm_set_profiling(0)
        assert p1 >= 1, p1
        assert joins == (1 if split else 0), joins
        assert scans == 0, scans                       # no batch went through the LDS scan / HBM table
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
        for bt in batches:
            if pred:
                bt = O.filter_batch(bt, O.cmp_mask(bt.column(1), O.GT, 64.0))
            o.next(bt)
        util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"{ncols} input columns nulls={nulls} pred={pred}")


@pytest.mark.parametrize("scenario", ["three_wide_high_cardinality", "nulls_and_floats", "dictionary_grows", "eight_columns",
                                      "merged_afterwards", "few_groups"])
def test_tuple_dictionary_for_keys_beyond_one_word(scenario, monkeypatch):
    """Key sets that do not fit 63 bits even as per-column dictionary codes (several wide columns with many distinct values
    EACH) used to aggregate with one HBM atomic per row and accumulator word (agg_wide_kernel).  Round 3: the wide-key table is
    a DICTIONARY tuple -> group id (tuple_gid_kernel), the ids go through the single-key operator and the stored tuples are
    the result's keys.  Several batches, NULL keys, float keys (-0.0 / NaN identity as in the wide table), a predicate, a
    dictionary that has to grow between and inside batches, a program wide enough to be split behind it, foreign partial
    state merged in afterwards; bit-exact against the oracle."""
    import ctypes
    from oracle import oracle as O
    from vinum_amd import _lib as L
    rng = np.random.default_rng(len(scenario))
    if scenario == "three_wide_high_cardinality":     # 2.4e6 distinct values per column: 3 x (23 + 1) bits of dictionary codes
        n, groups = 3_000_000, 2_400_000
    else:                                             # the small cases switch the per-column dictionaries off to get here
        n, groups = 900_000, (40 if scenario == "few_groups" else 150_000)
        monkeypatch.setenv("VNM_AGG_NO_DICT", "1")
    g = rng.integers(0, groups, n)
    mix = lambda x, m: (x * m) % (2**62) - 2**61             # noqa: E731  (wide, distinct per group)
    cols = {"a": pa.array(mix(g.astype(np.int64), 977_000_003)), "b": pa.array(mix(g.astype(np.int64) ^ 0x5DEECE66D, 1_000_003)),
            "c": pa.array(g.astype(np.int64) * (1 << 33) - 99)}
    if scenario == "nulls_and_floats":
        f = (g % 1000).astype(np.float64) * 1e300 * np.where(g % 2 == 0, 1.0, -1.0)
        f[g % 997 == 0] = np.nan
        f[g % 991 == 0] = -0.0
        f[g % 983 == 0] = 0.0
        cols["c"] = pa.array(f)
        cols["a"] = pa.array(cols["a"].to_numpy(), mask=(g % 13 == 0))
        cols["b"] = pa.array(cols["b"].to_numpy(), mask=(g % 17 == 0))
    keys = ["a", "b", "c"]
    ncol = 8 if scenario == "eight_columns" else 2
    funcs = []
    kinds = [O.SUM, O.MAX, O.AVG, O.MIN, O.COUNT, O.SUM, O.AVG, O.MAX]
    for c in range(ncol):
        vals = rng.integers(0, 2**14, n).astype(np.float64) / 64.0 if c % 2 == 0 else rng.integers(-2**40, 2**40, n).astype(np.int64)
        cols[f"c{c}"] = pa.array(vals, mask=(rng.random(n) < 0.05) if c == 1 else None)
        funcs.append((kinds[c], f"c{c}", f"f{c}"))
    funcs.append((O.COUNT_STAR, "", "n"))
    t = pa.table(cols)
    if scenario == "dictionary_grows":       # a tiny first batch sizes the dictionary; the later ones bring the groups
        batches = [t.slice(0, 2000).to_batches()[0]] + util.sliced_batches(t.slice(2000), 300_000)
    else:
        batches = util.sliced_batches(t, n // 3)

    def launches(name):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        L.lib().vnm_profile_query(name, ctypes.byref(ms), ctypes.byref(cnt))
        return cnt.value
    for pred in ((None,) if scenario in ("merged_afterwards", "eight_columns") else (None, ("c0", ">", 100.0))):
        o = O.OracleAggregate(O.MULTI, keys, keys, funcs)
        for bt in batches:
            o.next(O.filter_batch(bt, O.cmp_mask(bt.column(3), O.GT, 100.0)) if pred else bt)
        L.lib().vnm_set_profiling(1)
        if scenario == "merged_afterwards":
            from vinum_amd import ops
            from vinum_amd.device import DeviceColumn
            names = t.schema.names
            fspec = [(f, names.index(col) if col else None, t.schema.field(col).type if col else None) for f, col, _ in funcs]
            halves = []
            for part in (batches[:2], batches[2:]):
                agg = ops.DeviceAggregate(O.MULTI, [pa.int64()] * 3, fspec, expected_groups=groups)
                for b in part:
                    agg.next([DeviceColumn.from_arrow(b.column(j)) for j in range(3)],
                             [DeviceColumn.from_arrow(b.column(names.index(col))) if col else None for _, col, _ in funcs], nrows=b.num_rows)
                halves.append(agg)
            a, b2 = halves
            nb = b2.finish()
            kw, aw = b2.dense_ptrs()
            a.merge(nb, kw, aw)            # a leaves tuple mode: its groups go to its own wide-key table first
            got = a.result_arrays([0, 1, 2], keys, [f[2] for f in funcs])
            a.close(); b2.close()
        else:
            got = gpu_aggregate(O.MULTI, keys, keys, funcs, batches, predicate=pred,
                                expected_groups=0 if scenario in ("dictionary_grows", "few_groups") else groups)
        ids, packs = launches(b"agg_tuple_ids"), launches(b"agg_pack_keys")
        L.lib().vnm_set_profiling(0)
        assert ids >= len(batches) and packs == 0, (ids, packs)
        util.assert_agg_equal(got, o.result(), funcs, keys, what=f"tuple dictionary {scenario} pred={pred}")


@pytest.mark.parametrize("shape", ["split_program", "tuple_dictionary"])
def test_split_and_tuple_paths_with_no_surviving_row(shape, monkeypatch):
    """A predicate nothing passes (and one that a single row passes) under the split program and under the tuple dictionary:
    zero groups / one group, through both result routes."""
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    n = 600_000
    g = rng.integers(0, 50_000, n).astype(np.int64)
    cols, keys, kind = {}, ["k"], O.SINGLE
    if shape == "tuple_dictionary":
        monkeypatch.setenv("VNM_AGG_NO_DICT", "1")
        cols = {"a": pa.array(g * 977_000_003 - 2**61), "b": pa.array((g ^ 0x5DEECE66D) * 1_000_003), "c": pa.array(g * 2**33)}
        keys, kind = ["a", "b", "c"], O.MULTI
    else:
        cols["k"] = pa.array(g * 31 - 7)
    ncol = 8 if shape == "split_program" else 2
    funcs = []
    for c in range(ncol):
        vals = rng.integers(0, 2**14, n).astype(np.float64) / 64.0
        if c == 0:
            vals[123_456] = 1e6                      # the one row the second predicate lets through
        cols[f"c{c}"] = pa.array(vals)
        funcs.append(((O.SUM, O.MAX, O.AVG, O.MIN)[c % 4], f"c{c}", f"f{c}"))
    funcs.append((O.COUNT_STAR, "", "n"))
    t = pa.table(cols)
    batches = util.sliced_batches(t, n // 2)
    for thr, expect in ((1e9, 0), (5e5, 1)):
        got = gpu_aggregate(kind, keys, keys, funcs, batches, predicate=("c0", ">", thr), expected_groups=50_000)
        assert got.num_rows == expect
        o = O.OracleAggregate(kind, keys, keys, funcs)
        for bt in batches:
            o.next(O.filter_batch(bt, O.cmp_mask(bt.column(len(keys)), O.GT, thr)))
        util.assert_agg_equal(got, o.result(), funcs, keys, what=f"{shape}: {expect} surviving row(s)")


@pytest.mark.parametrize("scenario", ["hintless", "two_keys_packed", "merged_afterwards", "narrow_key"])
def test_split_program_scenarios(scenario):
    """The program split (eight input columns) where the operator has to find out for itself that the groups are many (no
    hint: estimate on the first batch), under packed composite keys and a packed int32 key (the single-key operator inside
    splits), and when foreign partial state is merged into the operator afterwards (the parts are joined first)."""
    import ctypes
    from oracle import oracle as O
    from vinum_amd import _lib as L
    rng = np.random.default_rng(len(scenario))
    n = (1 << 22) + 12345 if scenario in ("hintless", "narrow_key") else 1_200_000
    groups = 40_000
    g = rng.integers(0, groups, n)
    cols = {}
    if scenario == "two_keys_packed":
        cols["a"] = pa.array((g % 200).astype(np.int64) - 7)
        cols["b"] = pa.array((g // 200).astype(np.int32))
        keys, kind = ["a", "b"], O.MULTI
    elif scenario == "narrow_key":
        cols["k"] = pa.array(g.astype(np.int32) - 5)
        keys, kind = ["k"], O.SINGLE
    else:
        cols["k"] = pa.array(g.astype(np.int64) * 31 - 5)
        keys, kind = ["k"], O.SINGLE
    funcs = []
    kinds = [O.SUM, O.MAX, O.AVG, O.MIN, O.COUNT, O.SUM, O.AVG, O.MAX]
    for c in range(8):
        vals = rng.integers(0, 2**14, n).astype(np.float64) / 64.0 if c % 2 == 0 else rng.integers(-2**40, 2**40, n).astype(np.int64)
        cols[f"c{c}"] = pa.array(vals)
        funcs.append((kinds[c], f"c{c}", f"f{c}"))
    funcs.append((O.COUNT_STAR, "", "n"))
    t = pa.table(cols)
    batches = util.sliced_batches(t, n if scenario in ("hintless", "narrow_key") else n // 3)
    hint = 0 if scenario in ("hintless", "narrow_key") else groups

    def launches(name):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        L.lib().vnm_profile_query(name, ctypes.byref(ms), ctypes.byref(cnt))
        return cnt.value
    o = O.OracleAggregate(kind, keys, keys, funcs)
    for bt in batches:
        o.next(bt)
    L.lib().vnm_set_profiling(1)
    if scenario == "merged_afterwards":
        from vinum_amd import ops
        from vinum_amd.device import DeviceColumn
        schema = t.schema
        fspec = [(f, schema.names.index(col) if col else None, schema.field(col).type if col else None) for f, col, _ in funcs]
        halves = []
        for part in (batches[:2], batches[2:]):
            agg = ops.DeviceAggregate(kind, [pa.int64()], fspec, expected_groups=groups)
            for b in part:
                dk = DeviceColumn.from_arrow(b.column(0))
                agg.next([dk], [DeviceColumn.from_arrow(b.column(schema.names.index(col))) if col else None for _, col, _ in funcs], nrows=b.num_rows)
            halves.append(agg)
        a, b2 = halves
        nb = b2.finish()
        kw, aw = b2.dense_ptrs()
        a.merge(nb, kw, aw)      # a still holds its parts: they are joined, then b2's groups merged in
        got = a.result_arrays([0], keys, [f[2] for f in funcs])
        a.close(); b2.close()
    else:
        got = gpu_aggregate(kind, keys, keys, funcs, batches, expected_groups=hint)
    joins = launches(b"agg_split_join")
    L.lib().vnm_set_profiling(0)
    assert joins >= 1, joins
    util.assert_agg_equal(got, o.result(), funcs, keys, what=f"split program, {scenario}")


@pytest.mark.parametrize("scenario", ["two_wide_int64", "float_and_wide", "many_distinct", "new_values_later",
                                      "three_wide", "four_wide", "six_wide_no_fit", "wide_plus_dims", "table_overflow"])
def test_multi_key_dictionary_coded_fields(scenario):
    """Key columns whose RANGES do not fit the packed word but whose distinct values do (hashed ids, float64 keys): the widest
    columns are coded through per-column dictionaries (code = table slot) and the rest packs as before, so the single-key paths
    apply instead of the wide-key table.  Covered: the value that equals the table's EMPTY sentinel (int64 -1), NULL keys, NaN /
    -0.0 / 0.0 / inf as distinct bit patterns (array_iterators.h:239-248), dictionaries growing in later batches, and a
    dictionary that overflows (demotion to the wide-key table, results intact).  Bit-exact against the oracle."""
    import ctypes
    from oracle import oracle as O
    from vinum_amd import _lib as L
    rng = np.random.default_rng(len(scenario) * 7 + 1)
    n = 300_000
    expect_demote = False

    def wide_values(count, seed):
        vals = np.random.default_rng(seed).integers(-2**63, 2**63 - 1, count).astype(np.int64)
        vals[0] = -1                                   # bit pattern of the table's EMPTY word
        vals[1 % count] = np.iinfo(np.int64).min
        return vals
    if scenario == "two_wide_int64":
        a = pa.array(wide_values(30, 1)[rng.integers(0, 30, n)], mask=rng.random(n) < 0.03)
        b = pa.array(wide_values(20, 2)[rng.integers(0, 20, n)])
        keys = {"a": a, "b": b}
    elif scenario == "float_and_wide":
        fv = np.array([np.nan, -0.0, 0.0, np.inf, -np.inf, 1e300, -1e-300, 2.5, np.float64.fromhex("0x1.8p+1")] + list(rng.random(40) * 1e9))
        a = pa.array(fv[rng.integers(0, len(fv), n)], mask=rng.random(n) < 0.02)
        b = pa.array(wide_values(50, 3)[rng.integers(0, 50, n)], mask=rng.random(n) < 0.02)
        keys = {"a": a, "b": b}
    elif scenario == "many_distinct":
        a = pa.array(wide_values(120_000, 4)[rng.integers(0, 120_000, n)])
        b = pa.array(wide_values(3, 5)[rng.integers(0, 3, n)])
        keys = {"a": a, "b": b}
    elif scenario == "new_values_later":
        ia = np.concatenate([rng.integers(0, 40, n // 2), rng.integers(20, 4000, n - n // 2)])
        a = pa.array(wide_values(4000, 6)[ia])
        b = pa.array(wide_values(10, 7)[rng.integers(0, 10, n)])
        keys = {"a": a, "b": b}
    elif scenario in ("three_wide", "four_wide", "six_wide_no_fit"):   # 6 x (12 + 1) bits do not fit: the wide-key table takes it
        nk = {"three_wide": 3, "four_wide": 4, "six_wide_no_fit": 6}[scenario]
        keys = {c: pa.array(wide_values(6 + i, 8 + i)[rng.integers(0, 6 + i, n)], mask=rng.random(n) < 0.01) for i, c in enumerate("abcdef"[:nk])}
    elif scenario == "wide_plus_dims":
        keys = {"a": pa.array(rng.integers(0, 12, n).astype(np.int32)),
                "b": pa.array(wide_values(500, 11)[rng.integers(0, 500, n)]),
                "c": pa.array(rng.integers(-3, 3, n).astype(np.int8), mask=rng.random(n) < 0.1)}
    else:  # the second batch has more distinct values than the table of the first (2^20 slots) can take
        n = 1_700_000
        first = 200_000
        ia = np.concatenate([rng.integers(0, 10, first), np.arange(n - first)])
        a = pa.array(wide_values(n - first, 12)[ia])
        b = pa.array(wide_values(2, 13)[rng.integers(0, 2, n)])
        keys = {"a": a, "b": b}
        expect_demote = True
    v = pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)
    t = pa.table({**keys, "v": v})
    names = list(keys)
    funcs = [(O.SUM, "v", "s"), (O.COUNT_STAR, "", "n"), (O.MAX, "v", "hi")]
    if scenario == "table_overflow":
        batches = [t.slice(0, 200_000).to_batches()[0], t.slice(200_000).combine_chunks().to_batches()[0]]
    else:
        batches = util.sliced_batches(t, n // 2)

    def launches(name):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        L.lib().vnm_profile_query(name, ctypes.byref(ms), ctypes.byref(cnt))
        return cnt.value
    for pred in (None, ("v", ">", 64.0)):
        L.lib().vnm_set_profiling(1)
        got = gpu_aggregate(O.MULTI, names, names, funcs, batches, predicate=pred)
        packs, demotes, ids = launches(b"agg_pack_keys"), launches(b"agg_demote"), launches(b"agg_tuple_ids")
        L.lib().vnm_set_profiling(0)
        # what does not (or no longer) pack goes on through the tuple dictionary, not through the wide-key table's per-row atomics
        assert (ids >= 1) == (scenario == "six_wide_no_fit" or expect_demote), (ids, scenario)
        if scenario == "six_wide_no_fit":
            assert packs == 0 and demotes == 0, (packs, demotes)
        else:
            assert packs == len(batches), (packs, demotes)          # every batch went through the pack kernel ...
            assert demotes == (1 if expect_demote else 0), (packs, demotes)   # ... and stayed packed unless the dictionary overflowed
        o = O.OracleAggregate(O.MULTI, names, names, funcs)
        for bt in batches:
            if pred:
                bt = O.filter_batch(bt, O.cmp_mask(bt.column(len(names)), O.GT, 64.0))
            o.next(bt)
        util.assert_agg_equal(got, o.result(), funcs, names, what=f"dictionary-coded keys {scenario} pred={pred}")


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VNM_FUZZ_SEEDS", "100")))))
def test_random_plans_vs_oracle(seed, monkeypatch):
    """Seeded differential test over the whole dispatch space: random key columns (1-3, mixed widths, NULLs), random
    function lists over random typed inputs (NULLs, narrow types), random group counts, hints (right / absent),
    predicates on any column, one or two batches, skewed or uniform keys.  Whatever path the operator picks (scan
    kernels, narrow / wide partitioned entries, packed composite keys, spill, fallbacks) must equal the oracle."""
    from oracle import oracle as O
    rng = np.random.default_rng(1000 + seed)
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "1000")
    if rng.random() < 0.5:
        monkeypatch.setenv("VNM_AGG_PART_L1_MAX", "4")
    cols, key_names, in_names, funcs, n, groups, skew = util.random_agg_case(rng)
    nkeys = len(key_names)
    pred = None
    r = rng.random()
    if r < 0.3 and in_names:
        pred = (in_names[0], ">", 5 if pa.types.is_integer(cols[in_names[0]].type) else 5.0)
    elif r < 0.5:
        cols["p"] = pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)
        pred = ("p", "<=", 90.0)
    t = pa.table(cols)
    names = t.schema.names
    kind = O.SINGLE if nkeys == 1 else O.MULTI
    hint = groups if rng.random() < 0.5 else 0
    batches = t.to_batches() if rng.random() < 0.5 else util.sliced_batches(t, n // 2 + 1)
    got = gpu_aggregate(kind, key_names, key_names, funcs, batches, predicate=pred, expected_groups=hint)
    o = O.OracleAggregate(kind, key_names, key_names, funcs)
    fed = []
    for b in batches:
        if pred:
            op = {">": O.GT, "<=": O.LE}[pred[1]]
            b = O.filter_batch(b, O.cmp_mask(b.column(names.index(pred[0])), op, pred[2]))
        o.next(b)
        fed.append(b)
    util.assert_agg_equal(got, o.result(), funcs, key_names, source=fed,
                          what=f"seed {seed}: keys {[str(cols[k].type) for k in key_names]} inputs "
                               f"{[str(cols[v].type) for v in in_names]} G~{groups} skew={skew} hint={hint} pred={pred}")


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VNM_FUZZ_SEEDS", "24")))))
def test_random_plans_sharded_exchange_simulated(seed):
    """The N-GPU path with random queries, on one GPU: the rows are dealt to `world` operators (the ranks), every rank
    buckets its partial groups by owner (vnm_agg_bucket_by_owner), owner o merges bucket o of every rank
    (vnm_agg_merge_rows) -- what exchange_bucketed does over RCCL -- and the union of the owners' results must be
    the oracle's answer over all rows.  Any key layout (packed composite keys included), any accumulator program."""
    import torch
    from oracle import oracle as O
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    rng = np.random.default_rng(11000 + seed)
    cols, key_names, in_names, funcs, n, groups, skew = util.random_agg_case(rng)
    world = int(rng.choice([2, 3, 8]))
    t = pa.table(cols)
    names = t.schema.names
    kind = O.SINGLE if len(key_names) == 1 else O.MULTI
    key_types = [t.schema.field(c).type for c in key_names]
    fspec = [(f, names.index(col) if col else None, t.schema.field(col).type if col else None) for f, col, _ in funcs]
    bounds = np.linspace(0, n, world + 1).astype(int)
    ranks = []
    for r in range(world):
        b = t.slice(int(bounds[r]), int(bounds[r + 1] - bounds[r])).combine_chunks().to_batches()
        agg = ops.DeviceAggregate(kind, key_types, fspec, expected_groups=groups if rng.random() < 0.5 else 0)
        for bb in b:
            dev = {c: DeviceColumn.from_arrow(bb.column(names.index(c))) for c in set(key_names) | set(in_names)}
            agg.next([dev[c] for c in key_names], [dev[col] if col else None for _, col, _ in funcs], nrows=bb.num_rows)
        ng = agg.finish()
        kw, aw = agg.layout()
        rows = torch.empty((max(ng, 1), kw + aw), dtype=torch.int64, device="cuda")
        counts = agg.bucket_by_owner(world, rows.data_ptr())
        assert sum(counts) == ng
        ranks.append((rows, counts))
    parts = []
    for o in range(world):
        m = ops.DeviceAggregate(kind, key_types, fspec)
        for rows, counts in ranks:
            start = sum(counts[:o])
            part = rows[start:start + counts[o]].contiguous()
            if part.shape[0]:
                m.merge_rows(part.shape[0], part.data_ptr())
        parts.append(m.result_arrays(list(range(len(key_names))), key_names, [f[2] for f in funcs]))
    got = pa.Table.from_batches([p for p in parts if p.num_rows] or parts[:1]).combine_chunks()
    got = got.to_batches()[0] if got.num_rows else parts[0]
    o = O.OracleAggregate(kind, key_names, key_names, funcs)
    for b in t.to_batches():
        o.next(b)
    util.assert_agg_equal(got, o.result(), funcs, key_names, source=t,
                          what=f"seed {seed}: world {world} keys {[str(c) for c in key_types]} G~{groups}")


@pytest.mark.parametrize("scenario", ["g1", "g3", "g16", "g40", "few_then_many", "two_columns_g5"])
def test_few_groups_many_rows_key_copies(scenario):
    """Low-cardinality keys over many tiles per workgroup: the scan kernel gives every key up to eight copies in its
    LDS table (lane-spread slots) once the first tile showed a handful of groups, and drops the copies again when
    the table fills up (few_then_many).  Quantised values: every partial sum is exact, so the float columns are
    compared bit for bit (util.assert_agg_equal)."""
    import pandas as pd
    from oracle import oracle as O
    rng = np.random.default_rng(len(scenario))
    n = 7_000_000
    if scenario == "few_then_many":
        k = np.concatenate([rng.integers(0, 3, n // 2), rng.integers(0, 1800, n - n // 2)]).astype(np.int64)
    else:
        g = {"g1": 1, "g3": 3, "g16": 16, "g40": 40, "two_columns_g5": 5}[scenario]
        k = rng.integers(0, g, n).astype(np.int64) * 1_000_003 - 7
    a = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    b = rng.integers(-2**40, 2**40, n).astype(np.int64)
    t = pa.table({"k": pa.array(k), "a": pa.array(a), "b": pa.array(b)})
    if scenario == "two_columns_g5":
        funcs = [(O.SUM, "a", "sa"), (O.MAX, "b", "mb"), (O.MIN, "b", "nb"), (O.COUNT_STAR, "", "n"), (O.AVG, "a", "aa")]
    else:
        funcs = [(O.SUM, "a", "sa"), (O.MIN, "a", "na"), (O.MAX, "a", "ma"), (O.COUNT_STAR, "", "n"), (O.AVG, "a", "aa")]
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, t.to_batches(max_chunksize=n))
    df = pd.DataFrame({"k": k, "a": a, "b": b}).groupby("k", sort=True)
    exp = {"k": None}
    ref = df.agg(sa=("a", "sum"), na=("a", "min"), ma=("a", "max"), n=("a", "size"), aa=("a", "mean"),
                 mb=("b", "max"), nb=("b", "min")).reset_index()
    order = np.argsort(got.column(0).to_numpy(zero_copy_only=False), kind="stable")
    assert got.num_rows == len(ref)
    np.testing.assert_array_equal(got.column(0).to_numpy(zero_copy_only=False)[order], ref["k"].to_numpy())
    for i, (_, _, out) in enumerate(funcs):
        col = got.column(1 + i).to_numpy(zero_copy_only=False)[order]
        want = ref[out].to_numpy()
        if out == "aa":   # sum / count in float64, as the reference computes it
            want = ref["sa"].to_numpy() / ref["n"].to_numpy()
        np.testing.assert_array_equal(col, want.astype(col.dtype), err_msg=f"{scenario}: {out}")


@pytest.mark.parametrize("scenario", ["int32", "nullable_int64", "int16_nulls", "float32", "uint32_pred", "demote_on_later_batch",
                                      "sliced_odd_offset", "unpackable_range", "hintless_large_batch"])
def test_single_key_packed_for_the_partitioned_path(scenario):
    """A single key that the fast paths do not take as it is (narrow type, NULLs, odd Arrow offset) is packed into a
    plain 64-bit word when the group count is large or unknown, runs through the single-key machinery and is unpacked
    at the end (the same PackParams as the multi-column case); identity rules as in the reference (NULL is a group)."""
    from oracle import oracle as O
    rng = np.random.default_rng(len(scenario) * 7)
    n, G = 500_000, 60_000
    hint = G
    pred = None
    v = pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)
    x = pa.array(rng.integers(-5000, 5000, n).astype(np.int64), mask=rng.random(n) < 0.05)
    if scenario == "int32":
        k = pa.array(rng.integers(-G // 2, G // 2, n).astype(np.int32))
    elif scenario == "nullable_int64":
        k = pa.array(rng.integers(0, G, n).astype(np.int64) * 1_000_003, mask=rng.random(n) < 0.02)
    elif scenario == "int16_nulls":
        k = pa.array(rng.integers(-30000, 30000, n).astype(np.int16), mask=rng.random(n) < 0.01)
    elif scenario == "float32":
        k = pa.array((rng.integers(0, G, n) / 8.0).astype(np.float32))
    elif scenario == "uint32_pred":
        k = pa.array(rng.integers(0, G, n).astype(np.uint32) + np.uint32(4_000_000_000))
        pred = ("v", ">", 20.0)
    elif scenario == "demote_on_later_batch":   # the second batch leaves the range planned on the first
        k = pa.array(np.concatenate([rng.integers(0, G, n // 2), rng.integers(0, G, n - n // 2) * 2**40]).astype(np.int64),
                     mask=rng.random(n) < 0.02)
    elif scenario == "sliced_odd_offset":
        k = pa.array(rng.integers(0, G, n + 3).astype(np.int64)).slice(3)
    elif scenario == "unpackable_range":        # spans the whole int64 range: stays on the general path
        k = pa.array(rng.integers(-2**62, 2**62, n).astype(np.int64) * 2, mask=rng.random(n) < 0.02)
    else:                                       # no hint, batch above the estimator threshold
        n = 5_000_000
        hint = 0
        v = pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)
        x = pa.array(rng.integers(-5000, 5000, n).astype(np.int64), mask=rng.random(n) < 0.05)
        k = pa.array(rng.integers(0, 300_000, n).astype(np.int32), mask=rng.random(n) < 0.02)
    t = pa.table({"k": k, "v": v, "x": x})
    funcs = [(O.SUM, "v", "sv"), (O.AVG, "v", "av"), (O.COUNT_STAR, "", "n"), (O.MIN, "x", "mn"), (O.MAX, "x", "mx"),
             (O.COUNT, "x", "cx"), (O.SUM, "x", "sx")]
    batches = util.sliced_batches(t, (n + 1) // 2) if scenario != "hintless_large_batch" else t.to_batches(max_chunksize=n)
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred, expected_groups=hint)
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in batches:
        if pred:
            b = O.filter_batch(b, O.cmp_mask(b.column(1), O.GT, 20.0))
        o.next(b)
    util.assert_agg_equal(got, o.result(), funcs, ["k"], what=scenario)


@pytest.mark.parametrize("vtype", ["float64", "int64", "uint64", "count_only", "nullable_falls_back"])
@pytest.mark.parametrize("pred", ["none", "on_input", "on_other"])
def test_one_group_register_kernel(vtype, pred):
    """No GROUP BY (OneGroupAggregate, one_group_aggregate.cpp): every accumulator kind in registers over one plain
    8-byte column (agg_onegroup_hot_kernel), odd row counts, predicate on the input or on another float64 column;
    a nullable input takes the generic kernel.  Quantised floats: sums are exact, so everything is bit-exact."""
    from oracle import oracle as O
    if pred == "on_input" and vtype != "float64":
        pytest.skip("the fused predicate is a float64 comparison")
    rng = np.random.default_rng(len(vtype) * 3 + len(pred))
    n = 3_000_001
    p = pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)
    if vtype == "float64":
        v = pa.array(rng.integers(-2**13, 2**13, n).astype(np.float64) / 128.0)
    elif vtype == "uint64":
        v = pa.array(rng.integers(0, 2**64 - 1, n, dtype=np.uint64))          # sums need the 128-bit lanes
    elif vtype == "nullable_falls_back":
        v = pa.array(rng.integers(-2**40, 2**40, n).astype(np.int64), mask=rng.random(n) < 0.1)
    else:
        v = pa.array(rng.integers(-2**62, 2**62, n).astype(np.int64))
    t = pa.table({"v": v, "p": p})
    if vtype == "count_only":
        funcs = [(O.COUNT_STAR, "", "n")]
    else:
        funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.MIN, "v", "lo"), (O.MAX, "v", "hi"), (O.COUNT, "v", "c"), (O.COUNT_STAR, "", "n")]
    pcol = {"none": None, "on_input": "v", "on_other": "p"}[pred]
    thr = 0.0 if pred == "on_input" else 20.0
    batches = util.sliced_batches(t, 2_000_000)
    got = gpu_aggregate(O.ONE_GROUP, [], [], funcs, batches, predicate=(pcol, ">", thr) if pcol else None)
    o = O.OracleAggregate(O.ONE_GROUP, [], [], funcs)
    for b in batches:
        if pcol:
            b = O.filter_batch(b, O.cmp_mask(b.column(t.schema.names.index(pcol)), O.GT, thr))
        o.next(b)
    util.assert_agg_equal(got, o.result(), funcs, [], what=f"one group {vtype} pred={pred}")


@pytest.mark.parametrize("vtype", ["float64", "int64"])
@pytest.mark.parametrize("pred", ["none", "gt_on_input", "ne_on_input", "gt_on_other"])
@pytest.mark.parametrize("groups", [3, 900])
def test_scan_kernel_nullable_input(vtype, pred, groups):
    """One nullable 8-byte input column through agg_hot_kernel<VNULL>: a NULL input only counts for COUNT(*); a NULL in
    the predicate column compares like NaN (`!=` is true, everything else false), as the reference's NumPy
    comparison sees it.  Several tiles per workgroup (register rotation, key copies for 3 groups)."""
    from oracle import oracle as O
    if pred.endswith("on_input") and vtype != "float64":
        pytest.skip("the fused predicate is a float64 comparison")
    rng = np.random.default_rng(groups + len(pred) + len(vtype))
    n = 5_000_001
    kidx = rng.integers(0, groups, n)
    k = pa.array(kidx.astype(np.int64) * 977 - 5)
    mask = (rng.random(n) < 0.15) | (kidx == 1)   # every input of one group is NULL: the group must still come out
    if vtype == "float64":
        v = pa.array(rng.integers(-2**13, 2**13, n).astype(np.float64) / 128.0, mask=mask)
    else:
        v = pa.array(rng.integers(-2**50, 2**50, n).astype(np.int64), mask=mask)
    p = pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)
    t = pa.table({"k": k, "v": v, "p": p})
    funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.MIN, "v", "lo"), (O.MAX, "v", "hi"), (O.COUNT, "v", "c"), (O.COUNT_STAR, "", "n")]
    spec = {"none": None, "gt_on_input": ("v", ">", 1.5, O.GT), "ne_on_input": ("v", "!=", 1.5, O.NE), "gt_on_other": ("p", ">", 20.0, O.GT)}[pred]
    batches = util.sliced_batches(t, 3_000_000)   # even batch offsets: the 16-byte pair loads apply
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=spec[:3] if spec else None, expected_groups=groups)
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in batches:
        if spec:
            b = O.filter_batch(b, O.cmp_mask(b.column(t.schema.names.index(spec[0])), spec[3], spec[2]))
        o.next(b)
    util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"nullable {vtype} pred={pred} G={groups}")


@pytest.mark.parametrize("case", ["uniform", "negative_sorted", "uint64_high", "sample_misses", "second_batch_shifted", "heavy_key", "p1_7", "nonquantised"])
@pytest.mark.parametrize("hint", [0, 600_000])
@pytest.mark.parametrize("groups", [900_000, 40_000, 3_000], ids=["G9e5", "G4e4_split_final", "G3e3_lds_scan"])
def test_dense_key_partitioned_path(case, hint, groups, monkeypatch):
    """Dense-key path (vnm_agg_dense.inc): int64 / uint64 keys whose sampled range fits 29 bits travel as scrambled
    codes, the final pass direct-addresses its LDS accumulators.  Checked bit-exact against the oracle: negative and
    sorted keys (the scrambling must spread them), keys above 2^63, keys the sample never saw (they spill to the scan
    kernel), a later batch outside the code range, one key holding a third of the rows (region overflow -> spill)."""
    from oracle import oracle as O
    from vinum_amd import _lib as L
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    if case == "p1_7":
        monkeypatch.setenv("VNM_DENSE_P1", "3")
    # groups = 4e4: a range of 2^16 codes -> 16 final partitions, each split over many workgroups whose partial tables
    # dpart_merge_kernel adds up; groups = 3e3 (hint-less): no scatter, every workgroup scans rows into a whole-range LDS table
    rng = np.random.default_rng(len(case) + hint)
    n = 1_500_000
    base = rng.integers(0, groups, n).astype(np.int64)
    v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    kt = np.int64
    if case == "negative_sorted":
        k = np.sort(base) - groups // 2
    elif case == "uint64_high":
        k = base.astype(np.uint64) + np.uint64(2**63 + 12345)
        kt = np.uint64
    elif case == "sample_misses":
        k = base.copy()
        k[7::200_003] = 50_000_000          # far outside the sampled range, never on a sampled row stride
        k[11::190_001] = -3
    elif case == "heavy_key":
        k = base.copy()
        k[rng.random(n) < 0.33] = 4242
    else:
        k = base
    if case == "nonquantised":
        v = rng.lognormal(2.0, 1.0, n)
    cols = {"k": pa.array(k.astype(kt)), "v": pa.array(v)}
    t = pa.table(cols)
    batches = util.sliced_batches(t, 1_000_000)
    if case == "second_batch_shifted":
        k2 = (rng.integers(0, groups, 600_000) + 5_000_000).astype(np.int64)
        t2 = pa.table({"k": pa.array(k2), "v": pa.array(rng.integers(0, 2**14, 600_000).astype(np.float64) / 128.0)})
        batches = t.to_batches() + t2.to_batches()
    funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT_STAR, "", "n")]
    for pred in (("v", ">", 64.0), None):
        check_path = case == "uniform" and hint == 0
        if check_path:
            L.lib().vnm_set_profiling(1)    # (re)starts the kernel-span counters
        got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred, expected_groups=hint)
        if check_path:   # the path this parametrisation is about really ran
            def launches(name):
                ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
                L.lib().vnm_profile_query(name, ctypes.byref(ms), ctypes.byref(cnt))
                return cnt.value
            scan, p1, p2 = launches(b"agg_scan"), launches(b"agg_part_scatter1"), launches(b"agg_part_scatter2")
            L.lib().vnm_set_profiling(0)
            if groups == 3_000:
                assert scan == len(batches) and p1 == 0, (scan, p1, p2)          # direct-addressed LDS scan per batch
            else:
                # ONE scatter level (split / plain final pass).  A scan launch = spilled keys; the short second batch of the
                # 9e5-group case is below the dense path's rows-per-code bound and takes two hash levels
                assert p1 == len(batches) and p2 <= (1 if groups == 900_000 else 0), (scan, p1, p2)
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
        for b in batches:
            if pred:
                b = O.filter_batch(b, O.cmp_mask(b.column(1), O.GT, 64.0))
            o.next(b)
        exp = o.result()
        if case == "nonquantised":   # float sums: exactly rounded here, sequential in the oracle -> compare keys / counts bitwise
            util.assert_batches_equal(got.select(["k", "n"]), exp.select(["k", "n"]), key_names=["k"], what=f"dense {case}")
        else:
            util.assert_agg_equal(got, exp, funcs, ["k"], what=f"dense {case} hint={hint} pred={pred}")


@pytest.mark.parametrize("shape", ["dense_large_g", "few_groups", "one_group", "with_pred", "hash_large_g", "nullable_materialised",
                                   "int_columns_projected", "deep_materialised"])
def test_expression_inside_aggregate(shape, monkeypatch):
    """`SELECT k, sum(expr), avg(expr), count(*) ... GROUP BY k` with the expression handed to the aggregate
    (vnm_agg_set_input_expr): in the hot shape it is evaluated in registers inside the dense / hash partition pass, the
    LDS scan and the no-GROUP-BY kernel; otherwise the library materialises it with one fused projection pass.  Every
    route must give what the reference computes -- NumPy evaluates the expression node by node into a temporary column
    (planner.py:384-417), the aggregate sums it -- bit for bit (quantised inputs: exact products and sums)."""
    from oracle import oracle as O
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    from vinum_amd import _lib as L
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(len(shape))
    n = 1_300_001
    groups = {"dense_large_g": 700_000, "hash_large_g": 300_000, "few_groups": 5, "with_pred": 2000}.get(shape, 1000)
    k = rng.integers(0, groups, n).astype(np.int64)
    if shape == "hash_large_g":
        k = k * 1_000_003_337 - 99          # sparse 64-bit keys: not dense in their range -> hash-partitioned path
    q = lambda hi: rng.integers(0, hi, n).astype(np.float64) / 128.0
    total, tax, tip = q(128), q(64), q(32)
    expr = ("mul", ("mul", ("sub", 1, "total"), ("add", 2, "tax")), ("sub", 1, "tip"))      # test_query_results.py:436-443
    ref_expr = np.multiply(np.multiply(np.subtract(1, total), np.add(2, tax)), np.subtract(1, tip))
    cols = {"total": pa.array(total), "tax": pa.array(tax), "tip": pa.array(tip)}
    if shape == "nullable_materialised":
        mask = rng.random(n) < 0.05
        cols["tip"] = pa.array(tip, mask=mask)
        ref_expr = np.multiply(np.multiply(np.subtract(1, total), np.add(2, tax)), np.subtract(1, np.where(mask, np.nan, tip)))
    if shape == "int_columns_projected":
        ia, ib = rng.integers(-50, 50, n).astype(np.int64), rng.integers(1, 9, n).astype(np.int64)
        cols = {"ia": pa.array(ia), "ib": pa.array(ib)}
        expr = ("div", ("mul", "ia", "ib"), 4)                  # int64 columns: true division -> float64
        ref_expr = np.divide(np.multiply(ia, ib), 4)
    if shape == "deep_materialised":
        expr = ("add", "total", ("add", "tax", ("add", "tip", ("add", "total", ("mul", "tax", ("sub", "tip", ("neg", "total")))))))
        ref_expr = total + (tax + (tip + (total + (tax * (tip - (-total))))))
    names = list(cols)
    dev = {c: DeviceColumn.from_arrow(a) for c, a in cols.items()}
    kcol = DeviceColumn.from_numpy(k)
    one = shape == "one_group"
    agg = ops.DeviceAggregate(L.ONE_GROUP if one else L.SINGLE_NUMERICAL, [] if one else [pa.int64()],
                              [(L.SUM, 77, pa.float64()), (L.AVG, 77, pa.float64()), (L.COUNT_STAR, None, None)],
                              expected_groups=groups if shape == "hash_large_g" else 0)
    agg.set_input_expr(0, expr, names)
    pred = None
    if shape == "with_pred":
        agg.set_predicate(">", 0.25)
        pred = dev["tax"]
    half = 700_000
    for lo, hi in ((0, half), (half, n)):         # two batches (the second through table / run merging)
        sl = lambda c: c.slice(lo, hi - lo)
        agg.next([] if one else [sl(kcol)], [None, None, None], pred=sl(pred) if pred is not None else None, nrows=hi - lo,
                 expr_cols=[sl(dev[c]) for c in names])
    got = agg.result_arrays([] if one else [0], [] if one else ["k"], ["s", "a", "n"])
    agg.close()
    keep = np.ones(n, bool) if shape != "with_pred" else tax > 0.25
    t = pa.table({"k": pa.array(k[keep]), "e": pa.array(ref_expr[keep])})
    funcs = [(O.SUM, "e", "s"), (O.AVG, "e", "a"), (O.COUNT_STAR, "", "n")]
    o = O.OracleAggregate(O.ONE_GROUP if one else O.SINGLE, [] if one else ["k"], [] if one else ["k"], funcs)
    for b in t.to_batches():
        o.next(b)
    exp = o.result()
    if shape in ("nullable_materialised",):       # NaN inputs: sums are NaN on both sides; compare through the NaN-aware helper
        util.assert_agg_equal(got, exp, funcs, [] if one else ["k"], exact_float_inputs=("e",), what=shape)
    else:
        util.assert_agg_equal(got, exp, funcs, [] if one else ["k"], exact_float_inputs=("e",), what=shape)


@pytest.mark.parametrize("groups", [30_000, 150_000])
def test_streamed_batches_mid_cardinality(groups):
    """A stream of batches with a few 1e4 ... 1e5 groups (what TableReaderOperator makes of a big table): the final pass of
    the partitioned path splits partitions over several workgroups; from the second batch on the table is not empty, so
    the splits write partial groups (duplicate keys) into a run that is folded into the table -- ADVICE r01: those batches
    used to fall back to the scan kernel's flush storms.  Six batches, bit-exact against the oracle."""
    from oracle import oracle as O
    rng = np.random.default_rng(groups)
    n = 2_400_000
    k = rng.integers(0, groups, n).astype(np.int64) * 1_000_003 + 17
    v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    t = pa.table({"k": pa.array(k), "v": pa.array(v)})
    funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT_STAR, "", "n")]
    batches = util.sliced_batches(t, 400_000)
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=("v", ">", 20.0), expected_groups=groups)
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in batches:
        o.next(O.filter_batch(b, O.cmp_mask(b.column(1), O.GT, 20.0)))
    util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"streamed G={groups}")


@pytest.mark.parametrize("program", ["count_star", "minmax_f64", "sum_i64_count", "distinct", "min_u64_avg"])
@pytest.mark.parametrize("groups", [1_500_000, 40_000, 3_000], ids=["G1.5e6", "G4e4_split_final", "G3e3_lds_scan"])
@pytest.mark.parametrize("hint", [0, 2_000_000])
def test_dense_paths_generic_programs(program, groups, hint, monkeypatch):
    """The dense-key paths with generic accumulator programs (dgen_* kernels): COUNT(*)-only and DISTINCT travel as bare codes,
    MIN / MAX / integer sums / AVG of int64, uint64 and float64 inputs as raw 64-bit values; two scatter levels or one, plain
    or split final pass, whole-table LDS scan; predicate on another column; keys outside the sampled range spill."""
    from oracle import oracle as O
    from vinum_amd import _lib as L
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(groups + len(program) + hint)
    n = 2_400_000 if groups > 1_000_000 else 1_200_000
    k = rng.integers(0, groups, n).astype(np.int64) - groups // 3
    # a few keys far outside the range, on rows the range sample (row i * len / 2^18 of each batch) never looks at: they spill
    blen = n // 2 + 2
    for start in range(0, n, blen):
        ln = min(blen, n - start)
        m = min(ln, 1 << 18)
        unsampled = np.setdiff1d(np.arange(ln), (np.arange(m, dtype=np.int64) * ln) // m)
        k[start + unsampled[:: max(1, len(unsampled) // 7)][:7]] = 10**9 + start
    cols = {"k": pa.array(k)}
    if program == "minmax_f64":
        cols["v"] = pa.array(rng.normal(0, 100, n))
        funcs = [(O.MIN, "v", "lo"), (O.MAX, "v", "hi")]
    elif program == "sum_i64_count":
        cols["v"] = pa.array(rng.integers(-2**40, 2**40, n).astype(np.int64))
        funcs = [(O.SUM, "v", "s"), (O.COUNT, "v", "c"), (O.COUNT_STAR, "", "n")]
    elif program == "min_u64_avg":
        cols["v"] = pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2) + np.uint64(1))
        funcs = [(O.MIN, "v", "lo"), (O.AVG, "v", "a"), (O.MAX, "v", "hi")]
    elif program == "distinct":
        funcs = []
    else:
        funcs = [(O.COUNT_STAR, "", "n")]
    cols["p"] = pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)
    t = pa.table(cols)
    batches = util.sliced_batches(t, blen)
    names = t.schema.names
    for pred in (("p", ">", 20.0), None):
        L.lib().vnm_set_profiling(1)
        got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred, expected_groups=hint)
        def launches(name):
            ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
            L.lib().vnm_profile_query(name, ctypes.byref(ms), ctypes.byref(cnt))
            return cnt.value
        p1 = launches(b"agg_part_scatter1")
        L.lib().vnm_set_profiling(0)
        # the dense kernels really ran: a scatter level per batch, or none at all for the small range (min_u64_avg keeps five
        # words per slot in LDS -- min, max, two 32-bit sum lanes, count: 2048-slot scan tables do not hold 3000 groups, so it
        # scatters there as well)
        if hint == 0:
            assert p1 == (0 if groups == 3_000 and program != "min_u64_avg" else len(batches)), (program, groups, p1)
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
        for b in batches:
            if pred:
                b = O.filter_batch(b, O.cmp_mask(b.column(names.index("p")), O.GT, 20.0))
            o.next(b)
        util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"dense generic {program} G={groups} hint={hint} pred={pred}")


@pytest.mark.parametrize("program", ["sum_avg_f64", "minmax_f64_count", "sum_i64_count_star", "avg_u64"])
@pytest.mark.parametrize("groups", [1_500_000, 40_000, 3_000], ids=["G1.5e6", "G4e4_split_final", "G3e3_lds_scan"])
@pytest.mark.parametrize("pred", ["self", "other", "none"])
def test_dense_paths_nullable_value_column(program, groups, pred, monkeypatch):
    """The dense-key paths over ONE NULLABLE 8-byte input column (VN kernels): a NULL value travels as a flag next to the entry's
    code through one or two scatter levels (or is read from the bitmap by the LDS scan) -- the row counts for COUNT(*) and makes
    its group exist (SUM / AVG / MIN / MAX of an all-NULL group are NULL), nothing else; under `WHERE v > x` the NULL rows are
    dropped by the filter.  Sliced batches (odd bitmap offsets are not 16-byte aligned columns: those batches take the other
    paths, even ones the dense kernels); keys outside the sampled range spill as two lists (entries, keys of NULL-value rows);
    a group whose values are all NULL; hint-less.  Bit-exact against the oracle."""
    from oracle import oracle as O
    from vinum_amd import _lib as L
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(groups + len(program) * 7 + len(pred))
    n = 2_400_000 if groups > 1_000_000 else 1_200_000
    k = rng.integers(0, groups, n).astype(np.int64) - groups // 3
    blen = n // 2 + 2
    far = []
    for start in range(0, n, blen):
        ln = min(blen, n - start)
        m = min(ln, 1 << 18)
        unsampled = np.setdiff1d(np.arange(ln), (np.arange(m, dtype=np.int64) * ln) // m)
        pick = start + unsampled[:: max(1, len(unsampled) // 9)][:9]
        k[pick] = 10**9 + start + (np.arange(len(pick)) % 3)
        far.extend(pick.tolist())
    null = rng.random(n) < 0.12
    null[far[::2]] = True                      # NULL-value rows among the keys that spill
    null[far[1::2]] = False
    null[k == (groups // 2 - groups // 3)] = True   # a group with nothing but NULLs
    if program in ("sum_avg_f64", "minmax_f64_count"):
        vals = rng.integers(-2**14, 2**14, n).astype(np.float64) / 128.0
        funcs = ([(O.SUM, "v", "s"), (O.AVG, "v", "a")] if program == "sum_avg_f64"
                 else [(O.MIN, "v", "lo"), (O.MAX, "v", "hi"), (O.COUNT, "v", "c")])
    elif program == "sum_i64_count_star":
        vals = rng.integers(-2**40, 2**40, n).astype(np.int64)
        funcs = [(O.SUM, "v", "s"), (O.COUNT, "v", "c"), (O.COUNT_STAR, "", "n")]
    else:
        vals = rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2) + np.uint64(1)
        funcs = [(O.AVG, "v", "a"), (O.COUNT_STAR, "", "n")]
    if pred == "self" and vals.dtype != np.float64:
        pytest.skip("the fused predicate compares float64 columns")
    cols = {"k": pa.array(k), "v": pa.array(vals, mask=null), "p": pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)}
    t = pa.table(cols)
    batches = util.sliced_batches(t, blen)
    predicate = {"self": ("v", ">", -20.0), "other": ("p", ">", 20.0), "none": None}[pred]
    L.lib().vnm_set_profiling(1)
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=predicate)
    ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
    L.lib().vnm_profile_query(b"agg_part_scatter1", ctypes.byref(ms), ctypes.byref(cnt))
    L.lib().vnm_set_profiling(0)
    if groups > 3_000:
        assert cnt.value >= 1, "the dense scatter did not run over the nullable column"
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    names = t.schema.names
    for b in batches:
        if predicate:
            b = O.filter_batch(b, O.cmp_mask(b.column(names.index(predicate[0])), O.GT, predicate[2]))
        o.next(b)
    util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"dense nullable {program} G={groups} pred={pred}")


@pytest.mark.parametrize("program", ["hot", "minmax_count_star", "nullable_sum"])
@pytest.mark.parametrize("groups", [1_500_000, 40_000], ids=["G1.5e6_two_levels", "G4e4_one_level"])
@pytest.mark.parametrize("power", [4.0, 12.0])
def test_dense_paths_skewed_keys(program, groups, power, monkeypatch):
    """Skewed keys on the dense path: k = floor(G u^p) (p = 4: the first key holds ~3 % of the rows and hundreds of keys far more
    than an even share; p = 12: ~30 % of the rows in one key).  The ring scatter gives a sub-tile two insert / flush rounds
    and then spills what is still pending with ONE reservation per workgroup (1 round once a workgroup has spilled); full
    regions spill whole blocks with one reservation per wave.  Hot shape (deferred final pass + spilled entries in the HBM table),
    a generic program, and a nullable value column (NULL flags in the spilled entries: the two-list route).  Two batches;
    bit-exact against the oracle."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(int(groups + power * 10 + len(program)))
    n = 2_400_000
    k = np.floor(groups * rng.random(n) ** power).astype(np.int64) + 5
    v = rng.integers(-2**14, 2**14, n).astype(np.float64) / 128.0
    mask = None
    if program == "hot":
        funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT_STAR, "", "n")]
    elif program == "minmax_count_star":
        funcs = [(O.MIN, "v", "lo"), (O.MAX, "v", "hi"), (O.COUNT_STAR, "", "n")]
    else:
        funcs = [(O.SUM, "v", "s"), (O.COUNT, "v", "c"), (O.COUNT_STAR, "", "n")]
        mask = rng.random(n) < 0.2
    t = pa.table({"k": pa.array(k), "v": pa.array(v, mask=mask)})
    batches = util.sliced_batches(t, n // 2)
    for predicate in (("v", ">", -40.0), None):
        got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=predicate)
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
        for b in batches:
            if predicate:
                b = O.filter_batch(b, O.cmp_mask(b.column(1), O.GT, predicate[2]))
            o.next(b)
        util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"skewed {program} G={groups} p={power} pred={predicate}")


@pytest.mark.parametrize("values", ["quantised", "lognormal"])
def test_stream_table_of_small_range_batches(values, monkeypatch):
    """A stream of batches over a small dense key range (whole-table LDS scan per batch): the batches' per-workgroup tables are
    added into ONE table that lives across the stream (dscan_accumulate_kernel) and the groups are written once
    (flush_scan_pending).  Seven batches: four through the stream table (keys outside the sampled range spill on the way), a tiny
    one (general scan, the table becomes a run first), one over a SHIFTED range (another code map: flush, new table), one more of
    the first kind.  Bit-exact against the oracle; float sums = math.fsum whatever the batch order."""
    import math
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(5 + len(values))
    G = 3000
    funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT_STAR, "", "n")]

    def batch(n, lo, far=0):
        k = rng.integers(0, G, n).astype(np.int64) + lo
        if far:
            m = min(n, 1 << 18)
            unsampled = np.setdiff1d(np.arange(n), (np.arange(m, dtype=np.int64) * n) // m)
            k[unsampled[:far]] = 10**12 + np.arange(far) % 3
        v = (rng.integers(0, 2**14, n).astype(np.float64) / 128.0) if values == "quantised" else rng.lognormal(3.0, 2.0, n) * rng.choice([-1.0, 1.0], n)
        return pa.record_batch({"k": pa.array(k), "v": pa.array(v)})
    batches = [batch(600_000, -1000, far=5), batch(500_000, -1000), batch(700_000, -1000, far=3), batch(600_000, -1000),
               batch(1_000, -1000), batch(600_000, 10_000_000), batch(600_000, -1000)]
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=("v", ">", -1e300))
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in batches:
        o.next(O.filter_batch(b, O.cmp_mask(b.column(1), O.GT, -1e300)))
    exp = o.result()
    if values == "quantised":
        util.assert_agg_equal(got, exp, funcs, ["k"], what="stream table")
    else:
        util.assert_agg_equal(got, exp, funcs, ["k"], exact_float_inputs=(), what="stream table", source=batches)
    # the sums are the correctly rounded exact sums, batch order or not
    allk = np.concatenate([b.column(0).to_numpy() for b in batches]); allv = np.concatenate([b.column(1).to_numpy() for b in batches])
    gk = got.column(0).to_numpy(zero_copy_only=False); gs = got.column(1).to_numpy(zero_copy_only=False)
    for key in (gk[0], gk[len(gk) // 2], gk[-1]):
        assert gs[list(gk).index(key)] == math.fsum(allv[allk == key]), key


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VNM_FUZZ_SEEDS", "96")))))
def test_random_hot_shape_paths_vs_oracle(seed, monkeypatch):
    """Seeded differential test of the north-star shape (int64 / uint64 key, float64 value, sum / avg / count, optional
    float64 predicate) across everything that decides its path: key range from a few hundred to millions of codes (direct-
    addressed LDS scan, split final pass, one / two scatter levels), sparse 64-bit keys (hash partitions), negative /
    sorted / clustered / heavy / outlying keys (spill), right, wrong and absent hints, one to three batches whose key ranges
    may drift, quantised and arbitrary values.  Keys, counts and quantised sums bit-exact; arbitrary float sums held to the
    exact-sum bound (util._assert_float_agg_exact)."""
    from oracle import oracle as O
    rng = np.random.default_rng(52_000 + seed)
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "50000")
    n = int(rng.integers(200_000, 900_000))
    groups = int(rng.choice([150, 700, 3_000, 7_500, 20_000, 120_000, 400_000, 2_500_000]))
    lo = int(rng.choice([0, -groups // 2, 10**12, -(2**62)]))
    pattern = str(rng.choice(["uniform", "sorted", "clustered", "heavy", "outliers", "sparse"]))
    base = rng.integers(0, groups, n).astype(np.int64)
    if pattern == "sorted":
        base = np.sort(base)
    elif pattern == "clustered":
        base = (base // 64) * 64 + (np.arange(n) % 3)
    elif pattern == "heavy":
        base[rng.random(n) < 0.3] = groups // 3
    k = base + lo
    if pattern == "outliers":
        k[rng.integers(0, n, 40)] = lo + groups * 50 + rng.integers(0, 1000, 40)
    elif pattern == "sparse":
        k = base * 1_000_003_337 + lo
    unsigned = lo >= 0 and rng.random() < 0.3
    karr = pa.array(k.astype(np.uint64) if unsigned else k)
    quantised = rng.random() < 0.7
    v = rng.integers(-2**13, 2**13, n).astype(np.float64) / 128.0 if quantised else rng.lognormal(1.0, 1.5, n) * rng.choice([-1.0, 1.0], n)
    if rng.random() < 0.25:      # infinities and NaNs: a group's sum is +-inf, or NaN once it holds a NaN or both infinities
        sp = rng.integers(0, n, 60)
        v[sp[:20]] = np.inf
        v[sp[20:40]] = -np.inf
        v[sp[40:]] = np.nan
    cols = {"k": karr, "v": pa.array(v)}
    pred = None
    r = rng.random()
    if r < 0.35:
        pred = ("v", ">", 3.0)
    elif r < 0.6:
        cols["p"] = pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)
        pred = ("p", "<=", 40.0)
    t = pa.table(cols)
    nb = int(rng.choice([1, 1, 2, 3]))
    batches = util.sliced_batches(t, (n // nb + 2) & ~1)     # even offsets: the 16-byte pair loads apply
    if nb > 1 and rng.random() < 0.4:                        # a later batch in a shifted key range
        m = 120_000
        k2 = rng.integers(0, groups, m).astype(np.int64) + lo + groups * 3
        c2 = {"k": pa.array(k2.astype(np.uint64) if unsigned else k2), "v": pa.array(rng.integers(0, 2**13, m).astype(np.float64) / 128.0)}
        if "p" in cols:
            c2["p"] = pa.array(rng.integers(0, 2**12, m).astype(np.float64) / 64.0)
        batches = batches + pa.table(c2).to_batches()
    hint = int(rng.choice([0, 0, groups, groups * 40, max(1, groups // 50)]))
    funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT, "v", "c"), (O.COUNT_STAR, "", "n")]
    funcs = [funcs[i] for i in sorted(rng.choice(4, size=int(rng.integers(1, 5)), replace=False))]
    if not any(f[1] for f in funcs):
        funcs.append((O.SUM, "v", "s"))
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred, expected_groups=hint)
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    names = t.schema.names
    fed = []
    for b in batches:
        if pred:
            op = {">": O.GT, "<=": O.LE}[pred[1]]
            b = O.filter_batch(b, O.cmp_mask(b.column(names.index(pred[0])), op, pred[2]))
        o.next(b)
        fed.append(b.select(["k", "v"]))
    what = f"seed {seed}: G~{groups} lo={lo} {pattern} unsigned={unsigned} quantised={quantised} hint={hint} pred={pred} batches={len(batches)}"
    # arbitrary floats: held to the exact-sum bound (1 ULP of fsum, never further from the reference than the reference is from exact)
    util.assert_agg_equal(got, o.result(), funcs, ["k"], exact_float_inputs=("v",) if quantised else (), what=what,
                          source=None if quantised else fed)


@pytest.mark.parametrize("groups", [50, 20_000, 700_000, 3_000_000])
def test_estimate_groups_entry(groups):
    """vnm_agg_estimate_groups (what ranks all_reduce before a multi-GPU step): the operator's own estimate of a batch's group
    count without aggregating it -- never below the truth (partitions are sized from it) and at most 1.3x the size of the key
    population it was drawn from (the uniform-occupancy model extrapolates to the population, plus a 15-20 % margin)."""
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    from vinum_amd import _lib as L
    rng = np.random.default_rng(groups)
    n = 4_000_000
    k = rng.integers(0, groups, n).astype(np.int64) * 7919 - 5
    truth = len(np.unique(k))
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.COUNT_STAR, None, None)])
    est = agg.estimate_groups(DeviceColumn.from_numpy(k), n)
    assert truth <= est <= 1.3 * min(groups, n) + 16, (groups, truth, est)
    # a key the estimator cannot read (int32) reports 0: callers then skip the agreement step
    agg32 = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int32()], [(L.COUNT_STAR, None, None)])
    assert agg32.estimate_groups(DeviceColumn.from_numpy(k.astype(np.int32)), n) == 0
    agg.close(); agg32.close()
