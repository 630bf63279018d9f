#!/usr/bin/env python3
"""Whole-query fixtures from the reference's OWN planner + executor (build container only).

SQL text cannot be parsed here (pglast, pinned 1.17 by setup.py:35, is absent and there is no network), so the Query
AST (vinum/parser/query.py:408-432) is built by hand from tests/golden/planner_cases.py and handed to the unchanged
QueryPlanner (vinum/planner/planner.py:330-507) and RecursiveExecutor (vinum/executor/executor.py).  `import vinum`
needs two modules the image lacks: `pglast` -> oracle/pglast_stub (import stub, parses nothing) and the pybind module
`vinum_lib` -> oracle/ref_vinum_lib.py (the same names over the REAL reference operators compiled into oracle/_ref).
Filter / projection arithmetic runs in the reference's own Python (NumPy + pyarrow.compute).  SURVEY.md §8(c) item 3.

Outputs (data only): planner_in.arrow is NOT written -- the input table is regenerated from its seed
(planner_cases.planner_table, SHA-256 in planner_cases.json); planner_<name>.arrow = the reference's result.

Usage:  PYTHONPATH=oracle/pglast_stub:/root/reference python -B tests/golden/gen_golden_planner.py
"""
import json
import os
import sys

import pyarrow as pa

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "pglast_stub"))
sys.path.append("/root/reference")
sys.dont_write_bytecode = True

from oracle import ref_vinum_lib  # noqa: E402

sys.modules["vinum_lib"] = ref_vinum_lib
import vinum  # noqa: E402,F401
from vinum.arrow.arrow_table import ArrowTable  # noqa: E402
from vinum.executor.executor import RecursiveExecutor  # noqa: E402
from vinum.parser.query import Column, Expression, Literal, Query, SortOrder, SQLExpression  # noqa: E402
from vinum.planner.planner import QueryPlanner  # noqa: E402

from tests.golden import planner_cases as P  # noqa: E402
from tests.golden.float_cases import table_digest  # noqa: E402

OPS = {"add": "ADDITION", "sub": "SUBTRACTION", "mul": "MULTIPLICATION", "div": "DIVISION", "mod": "MODULUS",
       "neg": "NEGATION", "bnot": "BINARY_NOT", "band": "BINARY_AND", "bor": "BINARY_OR", "bxor": "BINARY_XOR",
       "eq": "EQUALS", "ne": "NOT_EQUALS", "gt": "GREATER_THAN", "ge": "GREATER_THAN_OR_EQUAL", "lt": "LESS_THAN",
       "le": "LESS_THAN_OR_EQUAL", "and": "AND", "or": "OR", "not": "NOT", "is_null": "IS_NULL",
       "is_not_null": "IS_NOT_NULL", "in": "IN", "not_in": "NOT_IN", "between": "BETWEEN", "not_between": "NOT_BETWEEN"}


def to_ast(e, alias=None):
    if isinstance(e, str):
        return Column(e, alias)
    if isinstance(e, (int, float)):
        return Literal(e, alias)
    op, args = e[0], e[1:]
    if op == "fn":
        return Expression(SQLExpression.FUNCTION, tuple(to_ast(a) for a in args[1:]), function_name=args[0], alias=alias)
    if op in ("in", "not_in"):   # parser.py:151-160: the value list is ONE Literal holding a Python list
        return Expression(SQLExpression[OPS[op]], (to_ast(args[0]), Literal(list(args[1]))), alias=alias)
    return Expression(SQLExpression[OPS[op]], tuple(to_ast(a) for a in args), alias=alias)


def has_agg(e):
    return isinstance(e, list) and (e[0] == "fn" or any(has_agg(a) for a in e[1:]))


def run(case, table):
    sel = tuple(to_ast(e, a) for e, a in zip(case["select"], case["aliases"]))
    is_agg = bool(case["group_by"]) or any(has_agg(e) for e in case["select"])
    q = Query(table.schema, sel, is_agg, case["distinct"],
              to_ast(case["where"]) if case["where"] is not None else None,
              tuple(to_ast(e) for e in case["group_by"]),
              to_ast(case["having"]) if case["having"] is not None else None,
              tuple(to_ast(e) for e in case["order_by"]),
              tuple(SortOrder[s] for s in case["sort_order"]), case["limit"], case["offset"])
    plan = QueryPlanner(q, table=ArrowTable(table)).plan_query()
    return RecursiveExecutor().execute(plan).get_table()


def main():
    vinum.set_batch_size(6000)   # several batches per query (the reference default is 10 000)
    table = P.planner_table()
    tables = {"main": table, "null": P.null_table()}
    meta = {"table_sha256": table_digest(table), "null_table_sha256": table_digest(tables["null"]), "cases": {}, "pyarrow": pa.__version__,
            "generator": "tests/golden/gen_golden_planner.py: the reference's QueryPlanner + RecursiveExecutor over hand-built Query ASTs"}
    for case in P.CASES:
        out = run(case, tables[case.get("table", "main")])
        with pa.OSFile(os.path.join(HERE, f"planner_{case['name']}.arrow"), "wb") as f:
            with pa.ipc.new_file(f, out.schema) as w:
                w.write_table(out.combine_chunks())
        meta["cases"][case["name"]] = {"rows": out.num_rows, "columns": out.schema.names, "types": [str(t) for t in out.schema.types]}
        print(f"{case['name']:28s} {out.num_rows:6d} rows  {out.schema.names}")
    with open(os.path.join(HERE, "planner_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
