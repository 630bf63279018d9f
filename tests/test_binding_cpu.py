"""Seam B2 on the CPU: the adapter consumes what the reference's UNCHANGED planner builds.

Runs only where /root/reference exists (the build container): the reference's QueryPlanner plans every query of
tests/golden/planner_cases.py twice -- as shipped, and with vinum_amd.binding.install() applied (which only rebinds the
FilterOperator / ProjectOperator names in the planner's namespace).  The GPU operators it then instantiates must hold
exactly the expressions vinum_amd's own planner derives from the same query, i.e. the VectorizedExpression trees carrying
the reference's registry callables lower to the programs the GPU tests execute.  No GPU is touched."""
import os
import sys

import pytest

from tests.golden import planner_cases as P

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "vinum")) or
                                not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libvinum_ref.so")),
                                reason="needs the reference sources and oracle/_ref (build container only)")


@pytest.fixture(scope="module")
def ref_env():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "pglast_stub"))
    sys.path.append(REF)
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True           # /root/reference is read-only
    from oracle import ref_vinum_lib
    sys.modules["vinum_lib"] = ref_vinum_lib
    import vinum
    from tests.golden import gen_golden_planner as G
    yield vinum, G
    sys.dont_write_bytecode = old
    for m in [k for k in sys.modules if k == "vinum" or k.startswith("vinum.") or k in ("vinum_lib", "pglast", "pglast.enums")]:
        del sys.modules[m]
    sys.path.remove(REF)
    sys.path.remove(os.path.join(ROOT, "oracle", "pglast_stub"))


def _chain(op):
    out = []
    while op is not None:
        out.append(op)
        op = getattr(op, "_parent_operator", None)
    return out[::-1]


def test_lower_reads_the_reference_registry(ref_env):
    vinum, G = ref_env
    from vinum.core import expressions as E
    from vinum.core.base import VectorizedExpression as VE
    from vinum.core.aggregate import AggregateFunction as AF
    from vinum.core.functions import FunctionType as FT
    from vinum.parser.query import Column, Literal, SQLExpression
    from vinum_amd import binding as B
    from vinum_amd.planner import _t
    index = B.registry_index(E.EXPRESSION_FUNCTIONS)
    classes = (Column, Literal, VE, AF, SQLExpression, FT, E.BINARY_EXPRESSIONS)
    for case in P.CASES:
        for e in case["select"] + ([case["where"]] if case["where"] is not None else []):
            e = _t(e)
            if isinstance(e, tuple) and not B.ops.columns_of(e) and e[0] != "fn":
                continue
            if isinstance(e, tuple) and e[0] == "fn" and len(e) > 2 and not isinstance(e[2], str):
                continue     # aggregate over an expression: the planner replaces the argument by a column first
            tree = B.vectorize(e, registry=E.EXPRESSION_FUNCTIONS, classes=classes)   # the reference's own classes + callables
            assert B.lower(tree, index) == B.lower(B.vectorize(e)) == _norm(e), case["name"]


def _norm(e):
    if isinstance(e, tuple) and e and e[0] in ("in", "not_in"):
        return (e[0], _norm(e[1]), tuple(e[2]))
    if isinstance(e, tuple):
        return tuple([e[0]] + [_norm(x) for x in e[1:]])
    return e


@pytest.mark.parametrize("case", P.CASES, ids=lambda c: c["name"])
def test_installed_operators_hold_the_programs_our_planner_derives(case, ref_env):
    vinum, G = ref_env
    from vinum.arrow.arrow_table import ArrowTable
    from vinum.parser.query import Query, SortOrder
    from vinum.planner import planner as RP
    from vinum_amd import binding as B
    from vinum_amd import planner as OP
    table = P.TABLES[case.get("table", "main")]()
    saved = (RP.FilterOperator, RP.ProjectOperator)
    try:
        B.install(vinum)
        sel = tuple(G.to_ast(e, a) for e, a in zip(case["select"], case["aliases"]))
        is_agg = bool(case["group_by"]) or any(G.has_agg(e) for e in case["select"])
        q = Query(table.schema, sel, is_agg, case["distinct"],
                  G.to_ast(case["where"]) if case["where"] is not None else None,
                  tuple(G.to_ast(e) for e in case["group_by"]),
                  G.to_ast(case["having"]) if case["having"] is not None else None,
                  tuple(G.to_ast(e) for e in case["order_by"]),
                  tuple(SortOrder[s] for s in case["sort_order"]), case["limit"], case["offset"])
        plan = RP.QueryPlanner(q, table=ArrowTable(table)).plan_query()
    finally:
        RP.FilterOperator, RP.ProjectOperator = saved
    ops_ref = _chain(plan)
    filters = [o for o in ops_ref if isinstance(o, B.GpuFilterOperator)]
    projects = [o for o in ops_ref if isinstance(o, B.GpuProjectOperator)]
    steps = dict((s[0], s) for s in OP.plan_query(case, table).steps)
    # WHERE / HAVING: same predicate program (modulo the internal names of aggregate output columns)
    n_filters = (case["where"] is not None) + (case["having"] is not None)
    assert len(filters) == n_filters
    if case["where"] is not None:
        assert filters[0].predicate == _norm(OP._t(case["where"])), "WHERE lowered differently"
    if case["having"] is not None:
        assert _shape(filters[-1].predicate) == _shape(steps["having"][1]), "HAVING lowered differently"
    # the final projection: same expressions up to the names of intermediate columns, same output names
    final = projects[-1]
    assert list(final._inner._col_names) == list(steps["project"][2])
    assert [_shape(e) for e in final.expressions] == [_shape(e) for e in steps["project"][1]]
    # expressions inside aggregates / GROUP BY expressions: one keep-input projection holding the same expressions
    # (our planner keeps an aggregate ARGUMENT expression inside the aggregate function -- the kernel evaluates it -- and
    # only projects GROUP BY expressions; the reference projects both: the union must be the same set of expressions)
    ours = [e for e, _ in steps["project_inner"][1]] if "project_inner" in steps else []
    if "aggregate" in steps:
        ours += [arg for (_f, arg), _out in steps["aggregate"][2] if isinstance(arg, tuple)]
    if ours:
        inner = [p for p in projects[:-1] if p._inner._keep]
        assert inner, "the reference plan has no pre-aggregate projection"
        assert sorted(map(repr, (_norm(e) for e in inner[-1].expressions))) == sorted(repr(e) for e in ours)


def _shape(e):
    """expression with column names that are planner-internal ids replaced by '#' (aggregate outputs, shared ids)"""
    if isinstance(e, str):
        return "#" if (e.startswith("__") or any(ch.isdigit() for ch in e) and ("_" in e)) else e
    if isinstance(e, tuple):
        if e[0] in ("in", "not_in"):
            return (e[0], _shape(e[1]), tuple(e[2]))
        return tuple([e[0]] + [_shape(x) for x in e[1:]])
    return e
