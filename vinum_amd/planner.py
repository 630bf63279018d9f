"""Query planner for the GPU operators: the build's counterpart of QueryPlanner.plan_query
(vinum/planner/planner.py:330-507) for queries given as data instead of SQL text (the reference parses SQL with pglast,
which is out of scope -- SURVEY.md §2; its own tests feed Query ASTs of exactly this content).

    reader -> [Filter] -> [Project inner aggregate / group-by expressions, keep_input] -> [Aggregate] -> [Filter HAVING]
           -> [Sort] -> Project(select list) -> [Slice] -> Materialize                      (planner.py:360-505)

A query is a dict (see tests/golden/planner_cases.py):
    select     [expr]            aliases [str | None]      distinct bool
    where      expr | None       group_by [expr]           having expr | None
    order_by   [expr]            sort_order ["ASC" | "DESC"]   limit int | None   offset int
    expr := column name | int | float | (op, arg...)   with op as in vinum_amd.ops plus ("fn", name, arg...) for the
            aggregate functions count_star / count / sum / avg / min / max.

What the planner does with the aggregate part mirrors planner.py:380-469:
  * DISTINCT = GROUP BY every select expression (:380-382);
  * an aggregate over an EXPRESSION (`sum((1 - total) * (2 + tax))`, test_query_results.py:436-443) and a GROUP BY
    expression become columns of a pre-aggregate projection that keeps the input (:384-417);
  * every aggregate function node -- in SELECT, HAVING or ORDER BY -- becomes one function of the AggregateOperator and is
    replaced by its output column in the post-aggregate expressions (:419-448), so HAVING is a filter and
    `sum(a) / count(*)` a projection over the G-row result.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import pyarrow as pa

from .core import (AggregateFunction, AggregateOperator, FileReaderOperator, FilterOperator, MaterializeTableOperator,
                   ProjectOperator, SliceOperator, SortOperator, TableReaderOperator)
from .ops import columns_of

AGG_FUNCS = ("count_star", "count", "sum", "avg", "min", "max")     # vinum/core/functions.py:389-396


def _t(e):
    """JSON lists -> tuples (hashable, and what ops.compile_expr takes); IN value lists stay lists."""
    if isinstance(e, (list, tuple)) and e and isinstance(e[0], str):
        if e[0] in ("in", "not_in"):
            return (e[0], _t(e[1]), tuple(e[2]))
        return tuple([e[0]] + [_t(x) for x in e[1:]])
    return e


def _is_fn(e) -> bool:
    return isinstance(e, tuple) and e[0] == "fn"


def _has_fn(e) -> bool:
    return isinstance(e, tuple) and (e[0] == "fn" or any(_has_fn(x) for x in e[1:]))


def _map(e, f):
    """Rebuild e bottom-up, f(node) may replace a node (applied to tuples only, children first)."""
    if not isinstance(e, tuple):
        return e
    if e[0] in ("in", "not_in"):
        return f((e[0], _map(e[1], f), e[2]))
    return f(tuple([e[0]] + [_map(x, f) for x in e[1:]]))


def output_names(select: Sequence, aliases: Sequence[Optional[str]]) -> List[str]:
    """QueryPlanner._column_names (planner.py:299-328): alias, else the column / function name, else col_<n>; a repeated
    name gets _<k>."""
    names, seen, unnamed = [], {}, 0
    for e, a in zip(select, aliases):
        if a:
            name = a
        elif isinstance(e, str):
            name = e
        elif _is_fn(e):
            name = e[1]
        else:
            name = f"col_{unnamed}"
            unnamed += 1
        if name in seen:
            seen[name] += 1
            name = f"{name}_{seen[name]}"
        else:
            seen[name] = 0
        names.append(name)
    return names


class Plan:
    """The operator chain plus what was decided on the way (tests compare it with the reference's plan shape)."""

    def __init__(self, root, steps):
        self.root, self.steps = root, steps

    def execute(self) -> pa.Table:
        return next(self.root.next())


def plan_query(query: Dict, source, expected_groups: int = 0) -> Plan:
    select = [_t(e) for e in query["select"]]
    aliases = list(query.get("aliases") or [None] * len(select))
    where = _t(query.get("where"))
    group_by = [_t(e) for e in query.get("group_by") or []]
    having = _t(query.get("having"))
    order_by = [_t(e) for e in query.get("order_by") or []]
    sort_order = [1 if str(s).upper().endswith("DESC") else 0 for s in (query.get("sort_order") or ["ASC"] * len(order_by))]
    limit, offset = query.get("limit"), query.get("offset") or 0
    distinct = bool(query.get("distinct"))
    steps = []

    # ---- column pruning (planner.py:346-371): only what the query touches is staged into HBM
    used: List[str] = []
    for e in select + ([where] if where is not None else []) + group_by + ([having] if having is not None else []) + order_by:
        for c in ([e] if isinstance(e, str) else columns_of(_strip_fn(e))):
            if c not in used:
                used.append(c)
    if isinstance(source, pa.Table):
        if not used and source.num_columns:
            used = [source.schema.names[0]]          # count(*) only: keep one column for the row count (:354-355)
        op = TableReaderOperator(source, columns=used)
    else:
        if hasattr(source, "read_next_device_batch") and used:
            source._want = [c for c in used]          # column pruning reaches the CSV parser: only these fields are parsed
        op = FileReaderOperator(source, columns=used or None)
    steps.append(("read", tuple(used)))

    if where is not None:
        pred = _simple_predicate(where)
        op = FilterOperator(pred if pred is not None else where, op)
        steps.append(("filter", where))

    is_agg = distinct or bool(group_by) or any(_has_fn(e) for e in select)
    if distinct:
        group_by = group_by + [e for e in select if e not in group_by]      # planner.py:380-382
    if is_agg:
        inner: Dict[Tuple, str] = {}      # expression -> pre-aggregate column

        def inner_col(e):
            if isinstance(e, str):
                return e
            if e not in inner:
                inner[e] = f"__inner_{len(inner)}"
            return inner[e]

        # aggregate function nodes -> functions of the operator (one per distinct (function, input)).  An argument that is
        # an EXPRESSION stays an expression: the aggregate operator hands it to the kernel (evaluated in registers in the
        # hot shape) or projects it itself -- the reference plans a keep-input projection for it (planner.py:384-417)
        funcs: Dict[Tuple[str, object], str] = {}
        key_of_expr: Dict[Tuple, str] = {}

        def fn_col(node):
            if not _is_fn(node):
                return node
            name = node[1].lower()
            if name not in AGG_FUNCS:
                raise ValueError(f"unknown aggregate function {node[1]!r}")
            arg = node[2] if len(node) > 2 and not isinstance(node[2], (int, float)) else ""
            if isinstance(arg, tuple) and arg in key_of_expr:
                arg = key_of_expr[arg]                    # the argument is a GROUP BY expression: already a column
            if name == "count" and not arg:
                name = "count_star"
            key = (name, arg)
            if key not in funcs:
                funcs[key] = f"__agg_{len(funcs)}"
            return funcs[key]

        key_cols = [inner_col(e) for e in group_by]
        key_of = dict(zip(group_by, key_cols))
        key_of_expr.update({e: c for e, c in key_of.items() if isinstance(e, tuple)})

        def post(e):
            """an expression over the aggregate's OUTPUT: functions and group-by expressions become columns"""
            if e in key_of:
                return key_of[e]
            return _map(e, lambda n: key_of.get(n, fn_col(n)))

        select_post = [post(e) for e in select]
        having_post = post(having) if having is not None else None
        order_post = [post(e) for e in order_by]
        if inner:
            op = ProjectOperator(list(inner.keys()), op, col_names=list(inner.values()), keep_input_table=True)
            steps.append(("project_inner", tuple(inner.items())))
        agg_funcs = [AggregateFunction(f, (arg or None) if isinstance(arg, str) else None, out, expr=arg if isinstance(arg, tuple) else None)
                     for (f, arg), out in funcs.items()]
        op = AggregateOperator(op, key_cols, agg_funcs, key_cols, expected_groups=expected_groups)
        steps.append(("aggregate", tuple(key_cols), tuple(funcs.items())))
        select, having, order_by = select_post, having_post, order_post
        if having is not None:
            pred = _simple_predicate(having)
            op = FilterOperator(pred if pred is not None else having, op)
            steps.append(("having", having))

    names = output_names([_t(e) for e in query["select"]], aliases)
    if order_by:
        # ORDER BY expressions are evaluated into extra columns, sorted by, and dropped by the final projection
        # (SortOperator.next, algebra.py:159-177)
        sort_cols, extra = [], {}
        for e in order_by:
            if isinstance(e, str):
                sort_cols.append(e)
            else:
                extra.setdefault(e, f"__sort_{len(extra)}")
                sort_cols.append(extra[e])
        if extra:
            op = ProjectOperator(list(extra.keys()), op, col_names=list(extra.values()), keep_input_table=True)
        push = (limit + offset) if limit is not None else 0
        op = SortOperator(sort_cols, sort_order, op, limit=push)
        steps.append(("sort", tuple(sort_cols), tuple(sort_order)))
    op = ProjectOperator(select, op, col_names=names)
    steps.append(("project", tuple(select), tuple(names)))
    if limit is not None:
        op = SliceOperator(limit, offset, op)
        steps.append(("slice", limit, offset))
    return Plan(MaterializeTableOperator(op), steps)


def execute(query: Dict, source, expected_groups: int = 0) -> pa.Table:
    return plan_query(query, source, expected_groups).execute()


def _strip_fn(e):
    """the same expression with aggregate function nodes replaced by their argument (for column discovery)"""
    def f(n):
        if _is_fn(n):
            return n[2] if len(n) > 2 else 0
        return n
    return _map(e, f) if isinstance(e, tuple) else e


def _simple_predicate(e):
    """`column <op> literal` -> the (column, op, literal) triple FilterOperator fuses (vnm_filter_cmp / the aggregate scan)"""
    ops_ = {"eq": "==", "ne": "!=", "gt": ">", "ge": ">=", "lt": "<", "le": "<="}
    if (isinstance(e, tuple) and len(e) == 3 and e[0] in ops_ and isinstance(e[1], str)
            and isinstance(e[2], (int, float)) and not isinstance(e[2], bool)):
        return (e[1], ops_[e[0]], e[2])
    return None
