"""GPU parity through the reference-shaped boundary: vinum_amd.vinum_lib mirrors the pybind11 module `vinum_lib`
(Arrow RecordBatches in, Arrow RecordBatch out, via the C ABI's Arrow-level entry points)."""
import numpy as np
import pyarrow as pa
import pytest

from tests import util
from tests.golden import gtest_fixtures as G

pytestmark = pytest.mark.gpu
MAN = util.manifest()


def _lib():
    from vinum_amd import vinum_lib
    return vinum_lib


def _agg(kind, groupby, agg_cols, funcs):
    vl = _lib()
    defs = [vl.AggFuncDef(vl.AggFuncType(f), col, out) for f, col, out in funcs]
    if kind == 0:
        return vl.OneGroupAggregate(defs)
    cls = vl.SingleNumericalHashAggregate if kind == 1 else vl.MultiNumericalHashAggregate
    return cls(groupby, agg_cols, defs)


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_gtest_known_answers_through_vinum_lib(name):
    c = G.CASES[name]
    table = G.table_for(c)
    for kind in c["kinds"]:
        agg = _agg(kind, c["groupby"], c["agg_cols"], c["funcs"])
        for b in G.feed_batches(table):
            agg.next(b)
        res = G.sort_result(agg.result(), c["sort_cols"])
        for i, exp in enumerate(c["expected"]):
            util.assert_col_equal(res.column(i), exp, f"{name}[{kind}] col {i}", ulps=1)


@pytest.mark.parametrize("case", MAN["agg"], ids=lambda c: c["name"])
def test_reference_golden_through_vinum_lib(case):
    table = util.read_ipc(case["input"])
    expected = util.read_ipc(case["expected"])
    funcs = [tuple(f) for f in case["funcs"]]
    agg = _agg(case["kind"], case["groupby"], case["agg_cols"], funcs)
    for b in util.sliced_batches(table, case["chunk"]):
        agg.next(b)
    util.assert_agg_equal(agg.result(), expected, funcs, case["agg_cols"], what=case["name"])


@pytest.mark.parametrize("case", MAN["sort"], ids=lambda c: c["name"])
def test_sort_golden_through_vinum_lib(case):
    vl = _lib()
    table = util.read_ipc(case["input"])
    expected = util.read_ipc(case["expected"])
    s = vl.Sort(case["cols"], [vl.SortOrder(o) for o in case["orders"]])
    for b in util.sliced_batches(table, case["chunk"]):
        s.next(b)
    util.assert_batches_equal(s.sorted(), expected, what=case["name"])


def test_error_behaviour_matches_reference():
    vl = _lib()
    b = pa.RecordBatch.from_arrays([pa.array([1, 2]), pa.array(["a", "b"])], names=["k", "s"])
    agg = vl.SingleNumericalHashAggregate(["nope"], ["nope"], [vl.AggFuncDef(vl.COUNT_STAR, "", "n")])
    with pytest.raises(RuntimeError, match="Column not found: nope"):       # base_aggregate.cpp:121-131
        agg.next(b)
    agg = vl.SingleNumericalHashAggregate(["k"], ["k"], [vl.AggFuncDef(vl.SUM, "s", "x")])
    with pytest.raises(RuntimeError, match=r"not supported by sum\(\)"):     # agg_func_factory.cpp:174
        agg.next(b)
    with pytest.raises(RuntimeError):
        vl.GenericHashAggregate(["s"], ["s"], [])
    assert vl.import_pyarrow() == 0
    assert repr(vl.AggFuncDef(vl.SUM, "a", "b")) == "<AggFuncDef col_name: a, out_col_name: b>"


def test_table_batch_reader():
    vl = _lib()
    t = pa.table({"a": np.arange(10)})
    r = vl.TableBatchReader(t)
    r.set_batch_size(4)
    sizes = []
    while True:
        b = r.next()
        if b is None:
            break
        sizes.append(b.num_rows)
    assert sizes == [4, 4, 2]


def test_large_batches_through_pinned_staging():
    """Batches far above the 32 MiB pinned ring (and above the estimator threshold): Arrow buffers -> pinned
    double buffer -> HBM -> hint-less aggregate (estimated cardinality) -> Arrow result, vs the oracle."""
    from oracle import oracle as O
    vl = _lib()
    rng = np.random.default_rng(77)
    n = 6_000_000
    t = pa.table({"k": pa.array(rng.integers(0, 200_000, n).astype(np.int64)),
                  "v": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0),
                  "w": pa.array(rng.integers(-1000, 1000, n).astype(np.int64), mask=rng.random(n) < 0.1)})
    funcs = [(O.SUM, "v", "sv"), (O.AVG, "v", "av"), (O.COUNT_STAR, "", "n")]
    agg = _agg(1, ["k"], ["k"], funcs)
    for b in t.to_batches(max_chunksize=4_500_000):
        agg.next(b)
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in t.to_batches():
        o.next(b)
    util.assert_agg_equal(agg.result(), o.result(), funcs, ["k"], what="large batches")
    funcs2 = [(O.SUM, "w", "sw"), (O.MIN, "w", "mn"), (O.MAX, "v", "mx"), (O.COUNT, "w", "cw")]
    agg = _agg(1, ["k"], ["k"], funcs2)
    for b in t.to_batches(max_chunksize=4_500_000):
        agg.next(b)
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs2)
    for b in t.to_batches():
        o.next(b)
    util.assert_agg_equal(agg.result(), o.result(), funcs2, ["k"], what="large batches generic")


import os


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VNM_FUZZ_SEEDS", "30")))))
def test_random_aggregates_through_vinum_lib(seed):
    """Seeded differential test at the Arrow boundary: random aggregates (see util.random_agg_case) fed through the
    vinum_lib classes as record batches cut at random places, with non-zero Arrow offsets, vs the oracle."""
    from oracle import oracle as O
    rng = np.random.default_rng(3000 + seed)
    cols, key_names, in_names, funcs, n, groups, skew = util.random_agg_case(rng)
    t = pa.table(cols)
    off = int(rng.choice([0, 1, 7, 64, 1001]))
    t = t.slice(off)                                    # every column now carries an Arrow offset
    cuts = sorted(set(int(x) for x in rng.integers(1, max(t.num_rows - 1, 2), int(rng.integers(0, 4)))))
    bounds = [0] + cuts + [t.num_rows]
    batches = [t.slice(a, b - a).combine_chunks().to_batches()[0] for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    kind = O.SINGLE if len(key_names) == 1 else O.MULTI
    agg = _agg(kind, key_names, key_names, funcs)
    o = O.OracleAggregate(kind, key_names, key_names, funcs)
    for b in batches:
        agg.next(b)
        o.next(b)
    util.assert_agg_equal(agg.result(), o.result(), funcs, key_names,
                          what=f"seed {seed}: keys {[str(cols[k].type) for k in key_names]} inputs "
                               f"{[str(cols[v].type) for v in in_names]} G~{groups} off={off} cuts={cuts}")
