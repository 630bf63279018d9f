"""Pin the CPU oracle (oracle/vinum_oracle.c) before trusting it as the parity checker.

(1) the reference's own gtest known answers (tests/golden/gtest_fixtures.py),
(2) outputs of the REAL reference operators on seeded inputs (tests/golden/*.arrow, manifest.json),
(3) when oracle/_ref is present (build container), live comparison against the real reference.
All comparisons are bit-exact (the oracle follows the reference's evaluation order).
"""
import os

import numpy as np
import pyarrow as pa
import pytest

from oracle import oracle as O
from oracle import ref as R
from tests import util
from tests.golden import gtest_fixtures as G

MAN = util.manifest()
OPS = {"eq": O.EQ, "ne": O.NE, "gt": O.GT, "ge": O.GE, "lt": O.LT, "le": O.LE}


def _funcs(c):
    return [tuple(f) for f in c["funcs"]]


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_oracle_matches_gtest_known_answers(name):
    c = G.CASES[name]
    table = G.table_for(c)
    for kind in c["kinds"]:
        # non-numeric keys / string aggregate inputs: the pure-Python restatement around the C one (OracleGenericAggregate)
        non_numeric = kind == G.GENERIC or any(col and not O._is_numeric_type(table.schema.field(col).type) for _, col, _ in c["funcs"])
        agg = (O.OracleGenericAggregate if non_numeric else O.OracleAggregate)(kind, c["groupby"], c["agg_cols"], c["funcs"])
        for b in G.feed_batches(table):
            agg.next(b)
        res = G.sort_result(agg.result(), c["sort_cols"])
        assert res.num_columns == len(c["expected"])
        for i, exp in enumerate(c["expected"]):
            util.assert_col_equal(res.column(i), exp, f"{name}[{kind}] col {i}")


@pytest.mark.parametrize("case", MAN["agg"], ids=lambda c: c["name"])
def test_oracle_aggregate_matches_reference_golden(case):
    table = util.read_ipc(case["input"])
    expected = util.read_ipc(case["expected"])
    agg = O.OracleAggregate(case["kind"], case["groupby"], case["agg_cols"], _funcs(case))
    for b in util.sliced_batches(table, case["chunk"]):
        agg.next(b)
    util.assert_batches_equal(agg.result(), expected, key_names=case["agg_cols"], what=case["name"])


@pytest.mark.parametrize("case", MAN["sort_mixed"], ids=lambda c: c["name"])
def test_oracle_sort_with_non_numeric_columns_matches_reference_golden(case):
    """Sort over tables with string / binary / boolean / decimal columns, as keys and as payload (the reference's orderby_queries
    shapes, test_query_results.py:627-745, 1252-1266): the oracle against outputs of the reference's own Sort."""
    table = util.read_ipc(case["input"])
    expected = util.read_ipc(case["expected"])
    s = O.OracleSort(case["cols"], case["orders"])
    for b in util.sliced_batches(table, case["chunk"]):
        s.next(b)
    got = s.sorted()
    assert got.schema.equals(expected.schema), (got.schema, expected.schema)
    util.assert_batches_equal(got, expected, what=case["name"])   # every column, order-sensitive, bit-exact


@pytest.mark.parametrize("case", MAN["sort"], ids=lambda c: c["name"])
def test_oracle_sort_matches_reference_golden(case):
    table = util.read_ipc(case["input"])
    expected = util.read_ipc(case["expected"])
    s = O.OracleSort(case["cols"], case["orders"])
    for b in util.sliced_batches(table, case["chunk"]):
        s.next(b)
    util.assert_batches_equal(s.sorted(), expected, what=case["name"])  # order-sensitive


@pytest.mark.parametrize("case", MAN["filter"], ids=lambda c: c["name"])
def test_oracle_filter_matches_reference_golden(case):
    table = util.read_ipc(case["input"]).combine_chunks()
    exp_batches = util.read_ipc_batches(case["expected"])
    lit = float(case["literal"]) if case["literal_is_float"] else int(case["literal"])
    for (off, ln), exp in zip(case["slices"], exp_batches):
        batch = table.slice(off, ln).to_batches()[0]
        col = batch.column(batch.schema.get_field_index(case["column"]))
        mask = O.cmp_mask(col, OPS[case["op"]], lit)
        util.assert_batches_equal(O.filter_batch(batch, mask), exp, what=case["name"])


def test_oracle_filter_emit_null_matches_arrow():
    """Masks born from pc.and_/or_/is_null can carry NULLs: emit_null keeps those rows as NULL rows."""
    rng = np.random.default_rng(5)
    n = 1000
    batch = pa.RecordBatch.from_arrays(
        [pa.array(rng.normal(size=n), mask=rng.random(n) < 0.1), pa.array(rng.integers(0, 9, n))], names=["x", "y"])
    m = rng.random(n) < 0.5
    mv = rng.random(n) < 0.9
    amask = pa.array(m, mask=~mv)
    exp = batch.filter(amask, null_selection_behavior="emit_null")
    util.assert_batches_equal(O.filter_batch(batch, m, mv), exp, what="emit_null")


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_matches_live_reference(seed):
    rng = np.random.default_rng(100 + seed)
    n = 20000
    t = pa.table({
        "k": pa.array(rng.integers(-50, 50, n).astype(np.int64), mask=rng.random(n) < 0.02),
        "k2": pa.array(rng.integers(0, 5, n).astype(np.int16), mask=rng.random(n) < 0.02),
        "v": pa.array(rng.lognormal(2, 1, n), mask=rng.random(n) < 0.1),
        "w": pa.array(rng.integers(-2**62, 2**62, n).astype(np.int64), mask=rng.random(n) < 0.1),
    })
    funcs = [(O.COUNT_STAR, "", "n"), (O.COUNT, "v", "cv"), (O.SUM, "v", "sv"), (O.AVG, "v", "av"),
             (O.MIN, "v", "mnv"), (O.MAX, "v", "mxv"), (O.SUM, "w", "sw"), (O.AVG, "w", "aw")]
    for kind, gb in [(O.SINGLE, ["k"]), (O.MULTI, ["k", "k2"]), (O.ONE_GROUP, [])]:
        o = O.OracleAggregate(kind, gb, gb, funcs)
        r = R.RefAggregate(kind, gb, gb, funcs)
        for b in util.sliced_batches(t, 3000):
            o.next(b)
            r.next(b)
        util.assert_batches_equal(o.result(), r.result(), key_names=gb, what=f"live kind={kind}")


def test_oracle_float_sum_order_and_minmax_rule_match_the_reference():
    """Sequential float64 `sum += x` in row order (agg_funcs.h:294-305) and `(row < last) ^ is_max` (agg_funcs.h:198)
    on NaNs / mixed zeros: the oracle must reproduce the real reference bit for bit on the seeded float cases
    (tests/golden/fsum_ref.arrow, minmax_ref.arrow: outputs of oracle/_ref, inputs regenerated from the seeds)."""
    import json
    import os
    from tests.golden import float_cases as C
    with open(os.path.join(util.GOLDEN, "float_cases.json")) as f:
        sums = json.load(f)
    for table, funcs, chunk, digest, out in ((C.fsum_table(), C.FSUM_FUNCS, C.FSUM_CHUNK, sums["fsum_sha256"], "fsum_ref.arrow"),
                                             (C.minmax_table(), C.MINMAX_FUNCS, C.MINMAX_CHUNK, sums["minmax_sha256"], "minmax_ref.arrow")):
        assert C.table_digest(table) == digest
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
        for b in util.sliced_batches(table, chunk):
            o.next(b)
        util.assert_batches_equal(o.result(), util.read_ipc(out), key_names=["k"], what=out)


def test_oracle_under_asan_ubsan():
    """The C restatement (oracle/vinum_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer: the golden aggregate,
    sort and filter cases and the gtest known answers are replayed in a child process that preloads libasan and loads an
    instrumented build of the oracle.  A heap overflow, use after free or signed overflow in the oracle would make every
    parity test built on it meaningless."""
    import subprocess
    import sys
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan here")
    env = dict(os.environ, VNM_ORACLE_SANITIZE="1", LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.abspath(__file__),
                        "-k", "gtest_known_answers or reference_golden or emit_null"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
