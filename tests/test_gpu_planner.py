"""Whole queries on the GPU operators vs fixtures produced by the reference's OWN planner + executor.

tests/golden/planner_<case>.arrow were written in the build container by tests/golden/gen_golden_planner.py: the
reference's unchanged QueryPlanner / RecursiveExecutor over hand-built Query ASTs (pglast is absent)
with its C++ aggregate / sort operators compiled from the reference sources (oracle/_ref).  Here the same queries
(tests/golden/planner_cases.py) go through vinum_amd.planner -> the GPU operators: WHERE trees, projections, aggregates
over expressions, HAVING, post-aggregate arithmetic, GROUP BY expressions, DISTINCT, ORDER BY expressions, LIMIT / OFFSET.
Bit-exact (the inputs are quantised, so float sums are exact in any order), row order compared only where the query
orders."""
import json
import os

import numpy as np
import pyarrow as pa
import pytest

from tests import util
from tests.golden import planner_cases as P
from tests.golden.float_cases import table_digest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def table():
    t = P.planner_table()
    with open(os.path.join(util.GOLDEN, "planner_cases.json")) as f:
        meta = json.load(f)
    assert table_digest(t) == meta["table_sha256"], "NumPy produced a different input table than the generator saw"
    assert table_digest(P.null_table()) == meta["null_table_sha256"]
    return t


def _batch(t: pa.Table) -> pa.RecordBatch:
    t = t.combine_chunks()
    return t.to_batches()[0] if t.num_rows else pa.RecordBatch.from_arrays([pa.array([], f.type) for f in t.schema], names=t.schema.names)


def _compare(got: pa.Table, exp: pa.Table, case):
    assert got.schema.names == exp.schema.names, f"{case['name']}: columns {got.schema.names} != {exp.schema.names}"
    assert got.num_rows == exp.num_rows, f"{case['name']}: {got.num_rows} rows, reference {exp.num_rows}"
    g, e = _batch(got), _batch(exp)
    if case["ordered"] and not case["order_by"]:
        util.assert_batches_equal(g, e, what=case["name"])                       # LIMIT over the input order
    elif case["order_by"]:
        # ORDER BY fixes the order of the sort keys; ties may come in any order: compare as sets, then the key order
        util.assert_batches_equal(g, e, key_names=g.schema.names, what=case["name"])
        if all(isinstance(k, str) and k in g.schema.names for k in case["order_by"]):
            for k in case["order_by"]:
                util.assert_col_equal(g.column(k), e.column(k), f"{case['name']}: order of {k}")
    else:
        util.assert_batches_equal(g, e, key_names=g.schema.names, what=case["name"])


@pytest.mark.parametrize("case", P.CASES, ids=lambda c: c["name"])
def test_query_matches_the_reference_planner(case, table):
    from vinum_amd import planner, set_batch_size
    set_batch_size(6000)          # several batches per query, as in the generator
    try:
        got = planner.execute(case, table if case.get("table", "main") == "main" else P.TABLES[case["table"]]())
    finally:
        set_batch_size(1 << 24)
    _compare(got, util.read_ipc(f"planner_{case['name']}.arrow"), case)


@pytest.mark.parametrize("case", [c for c in P.CASES if c["name"].startswith(("filter_", "project_")) and c["limit"] is None],
                         ids=lambda c: c["name"])
def test_b2_adapter_runs_vectorized_expression_trees(case, table):
    """Seam B2: the GPU Filter / Project operators constructed the way the planner constructs the reference's
    (FilterOperator(predicate: VectorizedExpression, parent), ProjectOperator(arguments, parent, col_names)) from trees
    of VectorizedExpression-shaped objects carrying the registry callables."""
    from vinum_amd import binding as B
    from vinum_amd.core import MaterializeTableOperator, TableReaderOperator
    from vinum_amd.planner import output_names, _t
    op = TableReaderOperator(table)
    if case["where"] is not None:
        op = B.GpuFilterOperator(B.vectorize(_t(case["where"])), op)
    sel = [_t(e) for e in case["select"]]
    op = B.GpuProjectOperator([B.vectorize(e) for e in sel], op, col_names=output_names(sel, case["aliases"]))
    got = next(MaterializeTableOperator(op).next())
    _compare(got, util.read_ipc(f"planner_{case['name']}.arrow"), case)


def test_record_batch_filter_replacement(table):
    """device_filter == RecordBatch.filter(mask, emit_null) of record_batch.py:85-90 for a mask given as a predicate tree."""
    from vinum_amd import binding as B
    from vinum_amd.core import DeviceRecordBatch
    b = _batch(table.select(["a", "b", "i"]))
    dev = DeviceRecordBatch.from_arrow(b)
    pred = ("and", ("gt", "a", 10), ("lt", ("add", "b", "i"), 40.5))
    got = B.device_filter(dev, B.vectorize(pred)).to_arrow()
    a = b.column("a").to_numpy(zero_copy_only=False)
    bb = b.column("b").to_numpy(zero_copy_only=False)         # NULL -> NaN (record_batch.py:112-118)
    i = b.column("i").to_numpy()
    mask = pa.array(np.logical_and(a > 10, (bb + i) < 40.5))
    util.assert_batches_equal(got, b.filter(mask, null_selection_behavior="emit_null"), what="device_filter")
