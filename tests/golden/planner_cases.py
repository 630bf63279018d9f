"""Whole-query cases for the planner-level fixtures (pure data + NumPy: no reference code).

Each case is a query in a small JSON-able form that BOTH sides can build from:
  * tests/golden/gen_golden_planner.py turns it into the reference's Query AST (vinum/parser/query.py:408-432) and runs
    the reference's own QueryPlanner + RecursiveExecutor over it (build container only) -> planner_<name>.arrow;
  * the GPU tests turn it into vinum_amd's mirror of the same AST and run it through the GPU operators.

expr := column name | int | float | [op, arg...]
  op: add sub mul div mod neg  band bor bxor bnot  eq ne gt ge lt le  and or not  is_null is_not_null
      in / not_in [x, [v...]]   between / not_between [x, lo, hi]   fn [name, arg...]  (count_star sum avg min max count)
"""
import numpy as np
import pyarrow as pa


def planner_table() -> pa.Table:
    rng = np.random.default_rng(314)
    n = 20_000
    q = lambda hi: rng.integers(0, hi, n).astype(np.float64) / 128.0       # quantised: products / sums stay exact
    return pa.table({
        "k": pa.array(rng.integers(0, 60, n).astype(np.int64) * 7 - 3),
        "g": pa.array(rng.integers(0, 7, n).astype(np.int32)),
        "total": pa.array(q(128)),                                          # [0, 1)
        "tax": pa.array(q(64)),
        "tip": pa.array(q(32), mask=rng.random(n) < 0.04),
        "a": pa.array(q(128 * 128)),                                        # [0, 128)
        "b": pa.array(rng.integers(-500, 500, n).astype(np.float64) / 4.0, mask=rng.random(n) < 0.05),
        "i": pa.array(rng.integers(-1000, 1000, n).astype(np.int64)),
        "j": pa.array(rng.integers(1, 50, n).astype(np.int64)),
    })


def null_table() -> pa.Table:
    """The reference's NULL / NaN test table (vinum/tests/conftest.py:76-103, `create_null_test_data`: a vector its own tests
    hold, transcribed as data) + `city_id`, an integer code of `city_from` (Berlin 0, Munich 1, San Francisco 2, NULL stays
    NULL), so that the query of test_query_results.py:1270-1301 also runs through the reference's NUMERIC aggregate
    operators, which is what oracle/_ref can build (generic_hash_aggregate.cpp does not compile against Arrow 25)."""
    nan = float("nan")
    rows = [(1, 1602127614, None, True, None, "Munich", 52.51, 13.66, "Joe", None),
            (2, 1602217613, "2020-10-09T04:26:53", True, "Munich", "Riva", 48.51, 12.3, None, 143.15),
            (3, 1602304012, "2020-10-10T04:26:52", False, None, "Naples", 44.89, 14.23, "Joseph", 33.40),
            (4, 1602390411, "2020-10-11T04:26:51", None, "San Francisco", "Naples", 42.89, 15.89, "Joseph", 53.1),
            (5, None, "2020-10-12T04:26:50", True, "Berlin", "Riva", 44.89, 14.23, None, nan),
            (6, 1602563209, "2020-10-13T04:26:49", None, "Munich", "Riva", 48.51, 12.3, "Jonas", None),
            (7, None, None, None, "Berlin", "Munich", 44.89, 14.23, "Joseph", 33.40),
            (8, 1602736007, "2020-10-15T04:26:47", None, "Berlin", "Munich", 52.51, 13.66, "Joe", nan)]
    names = ("id", "timestamp", "date", "is_vendor", "city_from", "city_to", "lat", "lng", "name", "total")
    cols = {n: [r[i] for r in rows] for i, n in enumerate(names)}
    code = {"Berlin": 0, "Munich": 1, "San Francisco": 2}
    t = pa.table({"id": pa.array(cols["id"], pa.int64()), "timestamp": pa.array(cols["timestamp"], pa.int64()),
                  "date": pa.array(cols["date"], pa.string()), "is_vendor": pa.array(cols["is_vendor"], pa.bool_()),
                  "city_from": pa.array(cols["city_from"], pa.string()), "city_to": pa.array(cols["city_to"], pa.string()),
                  "lat": pa.array(cols["lat"], pa.float64()), "lng": pa.array(cols["lng"], pa.float64()),
                  "name": pa.array(cols["name"], pa.string()),
                  "total": pa.array(cols["total"], pa.float64(), from_pandas=False),
                  "city_id": pa.array([code.get(c) for c in cols["city_from"]], pa.int64())})
    return t


# what the reference's test expects of `... group by city_from order by city_from` over that table
# (vinum/tests/test_query_results.py:1270-1301; the datetime() / from_timestamp() counts are left out: no such functions here)
NULL_TABLE_EXPECTED = {
    "city_from": ("Berlin", "Munich", "San Francisco", None),
    "cnt_all": (3, 2, 1, 2), "cnt_total": (3, 1, 1, 1), "cnt_name": (2, 1, 1, 2), "cnt_date_str": (2, 2, 1, 1), "cnt_bool": (1, 1, 0, 2),
    "min_total": (float("nan"), 143.15, 53.1, 33.4), "max_total": (float("nan"), 143.15, 53.1, 33.4),
    "avg_total": (float("nan"), 143.15, 53.1, 33.4), "sum_total": (float("nan"), 143.15, 53.1, 33.4),
}


def Q(name, select, aliases=None, distinct=False, where=None, group_by=(), having=None, order_by=(), sort_order=(),
      limit=None, offset=0, ordered=False, table="main"):
    return dict(name=name, select=list(select), aliases=list(aliases) if aliases else [None] * len(select),
                distinct=distinct, where=where, group_by=list(group_by), having=having, order_by=list(order_by),
                sort_order=list(sort_order), limit=limit, offset=offset, ordered=ordered or bool(order_by), table=table)


fn = lambda name, *args: ["fn", name, *args]

CASES = [
    # ---- filter (seam B2: predicates as the planner builds them, vinum/core/expressions.py:27-48)
    Q("filter_simple", ["a", "b"], where=["gt", "a", 64.0]),
    Q("filter_tree", ["k", "a", "b", "i"], where=["or", ["and", ["gt", "a", 10], ["le", "b", 3.5]], ["not", ["eq", "i", 5]]]),
    Q("filter_between_in", ["k", "a"], where=["and", ["between", "a", 10, 50.5], ["in", "k", [4, 11, 18, 410]]]),
    Q("filter_not_between_not_in", ["g", "i"], where=["and", ["not_between", "i", -900, 900], ["not_in", "g", [0, 6]]]),
    Q("filter_is_null", ["k", "b"], where=["is_null", "b"]),
    Q("filter_not_null_arith", ["a", "tip"], where=["and", ["is_not_null", "tip"], ["gt", ["mul", "a", "tip"], 4]]),
    Q("filter_nary_and", ["a"], where=["and", ["gt", "a", 1], ["lt", "a", 120], ["ne", "g", 3]]),
    # ---- projection (vinum/core/expressions.py:13-24)
    Q("project_arith", [["add", ["mul", "a", 2], 1], ["sub", "a", "b"], ["mod", "i", 7], ["neg", "i"], ["div", "i", 3],
                        ["mul", ["add", "a", "b"], ["sub", "a", "tax"]]], aliases=["e1", "e2", "e3", "e4", "e5", "e6"]),
    Q("project_bits_scalar", [["band", "i", 255], ["bor", "j", 1024], ["bxor", "i", "j"], ["bnot", "j"], ["add", 1, 2], "a"],
      aliases=["e1", "e2", "e3", "e4", "three", None]),
    Q("project_where_limit", ["k", ["div", "a", "j"]], aliases=[None, "ratio"], where=["ge", "g", 2], limit=77, offset=13,
      ordered=True),
    # ---- aggregates with expressions inside, HAVING, post-aggregate expressions (planner.py:380-469;
    #      the query shape of vinum/tests/test_query_results.py:436-443)
    Q("agg_inner_expr", ["k", fn("sum", ["mul", ["mul", ["sub", 1, "total"], ["add", 2, "tax"]], ["sub", 1, "tip"]]), fn("count_star")],
      aliases=[None, "s", "n"], group_by=["k"]),
    Q("agg_having", ["k", fn("sum", "a"), fn("avg", "b")], aliases=[None, "s", "m"], group_by=["k"],
      having=["gt", fn("sum", "a"), 21000.0]),
    Q("agg_where_order_limit", ["k", fn("count_star"), fn("avg", "a")], aliases=[None, "n", "m"], where=["gt", "a", 64.0],
      group_by=["k"], order_by=["k"], sort_order=["DESC"], limit=5),
    Q("agg_post_expr", ["g", ["div", fn("sum", "a"), fn("count_star")], ["sub", fn("max", "i"), fn("min", "i")]],
      aliases=[None, "mean", "range"], group_by=["g"]),
    Q("agg_groupby_expr", [["mod", "j", 5], fn("sum", "i"), fn("count", "b")], aliases=["bucket", "s", "c"],
      group_by=[["mod", "j", 5]]),
    Q("agg_multi_key", ["g", "k", fn("min", "a"), fn("max", "b")], aliases=[None, None, "lo", "hi"], group_by=["g", "k"]),
    Q("agg_one_group", [fn("count_star"), fn("sum", "a"), fn("min", "b"), fn("avg", ["add", "i", "j"])],
      aliases=["n", "s", "lo", "m"], where=["gt", "b", 0]),
    # ---- DISTINCT = group-by on every select expression (planner.py:380-382)
    Q("distinct_col", ["g"], distinct=True),
    Q("distinct_two", ["g", "j"], distinct=True),
    # ---- ORDER BY (column / expression), LIMIT / OFFSET (algebra.py:126-247)
    Q("order_expr_limit", ["a", "b"], order_by=[["mul", "a", "j"]], sort_order=["DESC"], limit=10),
    Q("order_two_keys", ["g", "i", "a"], where=["lt", "a", 2.0], order_by=["g", "i"], sort_order=["ASC", "DESC"], limit=40, offset=5),
    # ---- the reference's NULL / NaN vector (test_query_results.py:1270-1301) over the integer code of city_from: MIN / MAX of
    #      (NaN, 33.4, NaN) are NaN -- MinMaxFunc::Update is row-order dependent (agg_funcs.h:188-201)
    Q("null_table_minmax", ["city_id", fn("count_star"), fn("count", "total"), fn("min", "total"), fn("max", "total"),
                            fn("avg", "total"), fn("sum", "total")],
      aliases=[None, "cnt_all", "cnt_total", "min_total", "max_total", "avg_total", "sum_total"], group_by=["city_id"],
      order_by=["city_id"], sort_order=["ASC"], table="null"),
    Q("null_table_minmax_one_group", [fn("min", "total"), fn("max", "total"), fn("count", "total")], aliases=["mn", "mx", "c"], table="null"),
    Q("null_table_minmax_where", ["city_id", fn("min", "total"), fn("max", "total")], aliases=[None, "mn", "mx"], where=["gt", "id", 5],
      group_by=["city_id"], order_by=["city_id"], sort_order=["ASC"], table="null"),
]
TABLES = {"main": planner_table, "null": null_table}
