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
