"""AggregateOperator (mirror of vinum/core/aggregate.py): picks the operator kind from the group-by key types
(:96-104), streams batches into the device aggregate, yields one result batch."""
from typing import Optional, Sequence, Tuple

import pyarrow as pa

from .. import _lib as L
from .. import ops
from .base import DeviceRecordBatch, Operator
from .algebra import FilterOperator

_FUNCS = {"COUNT": L.COUNT, "COUNT_STAR": L.COUNT_STAR, "MIN": L.MIN, "MAX": L.MAX, "SUM": L.SUM, "AVG": L.AVG}


class AggregateFunction:
    """expr (optional): the function's argument is an EXPRESSION over input columns (prefix tuple, see vinum_amd.ops) --
    `sum((1 - total) * (2 + tax))`.  The reference's planner projects it into a temporary column before the aggregate
    (planner.py:384-417); here the operator hands it to the kernel, which evaluates it in registers when it can."""

    def __init__(self, func: str, column: Optional[str] = None, out_name: Optional[str] = None, expr=None):
        self.func = func.upper()
        self.expr = expr
        if self.func == "COUNT" and not column and expr is None:
            self.func = "COUNT_STAR"                       # parser.py:210-211
        self.column = column or ""
        self.out_name = out_name or (f"{self.func.lower()}_{self.column}" if self.column else self.func.lower())


def _is_numeric(t: pa.DataType) -> bool:                   # aggregate.py:63-66
    return pa.types.is_integer(t) or pa.types.is_floating(t) or pa.types.is_temporal(t)


class AggregateOperator(Operator):
    def __init__(self, parent_operator: Operator, group_by_columns: Sequence[str], agg_funcs: Sequence[AggregateFunction],
                 agg_cols: Sequence[str], expected_groups: int = 0):
        # Filter -> Aggregate fusion (planner.py:373-378 + 463-469 in one scan): a directly preceding
        # `column <op> literal` filter is folded into the aggregate kernel, no filtered batch is materialised
        self._fused_pred: Optional[Tuple[str, str, object]] = None
        if isinstance(parent_operator, FilterOperator) and FilterOperator.is_simple(parent_operator.predicate):
            self._fused_pred = parent_operator.predicate
            parent_operator = parent_operator._parent_operator
        super().__init__(parent_operator)
        self._group_by, self._funcs, self._agg_cols = list(group_by_columns), list(agg_funcs), list(agg_cols)
        self._expected_groups = expected_groups
        self._agg = None

    def _init(self, batch: DeviceRecordBatch):
        key_types = [batch.column(c).arrow_type for c in self._group_by]
        self._key_dicts = {c: batch.column(c).dictionary for c in self._group_by}
        for f in self._funcs:
            if f.column and batch.column(f.column).dictionary is not None and f.func != "COUNT":
                raise RuntimeError(f"{f.func.lower()}() over the non-numeric column {f.column!r} is not on the GPU path")
        kind = L.ONE_GROUP if not self._group_by else (L.SINGLE_NUMERICAL if len(self._group_by) == 1 else L.MULTI_NUMERICAL)
        names = batch.column_names
        spec = []
        # one expression input per operator goes to the kernel (float64 columns only: the expression's type must be float64);
        # any other expression is projected per batch by this operator, exactly what the reference's planner plans
        exprs = []
        for f in self._funcs:
            if f.expr is not None and f.expr not in exprs:
                exprs.append(f.expr)
        self._kernel_expr, self._kernel_expr_cols, self._projected = None, [], {}
        for e in exprs:
            cols = ops.columns_of(e)
            f64 = all(batch.column(c).arrow_type == pa.float64() and batch.column(c).dictionary is None for c in cols)
            if self._kernel_expr is None and cols and f64 and len(cols) <= 16:
                self._kernel_expr, self._kernel_expr_cols = e, cols
            else:
                self._projected[e] = f"__expr_{len(self._projected)}"
        if self._projected:
            batch = self._with_projected(batch)
            names = batch.column_names
        for f in self._funcs:
            if f.expr is not None and f.expr == self._kernel_expr:
                spec.append((_FUNCS[f.func], 10_000, pa.float64()))
            elif f.expr is not None:
                col = self._projected[f.expr]
                spec.append((_FUNCS[f.func], names.index(col), batch.column(col).arrow_type))
            elif f.column:
                spec.append((_FUNCS[f.func], names.index(f.column), batch.column(f.column).arrow_type))
            else:
                spec.append((_FUNCS[f.func], None, None))
        # stream mode (vnm_agg_set_async): the batches the parent yields -- TableReaderOperator / the CSV reader hand them over one
        # by one, base_aggregate.cpp:23-45 -- are recorded and go to the device as the segments of one launch where the path
        # takes segments (the hot shape); every other shape is processed per call as before
        self._agg = ops.DeviceAggregate(kind, key_types, spec, expected_groups=self._expected_groups, stream_mode=True)
        if self._fused_pred:
            self._agg.set_predicate(self._fused_pred[1], self._fused_pred[2])
        if self._kernel_expr is not None:
            first = next(i for i, f in enumerate(self._funcs) if f.expr is not None and f.expr == self._kernel_expr)
            self._agg.set_input_expr(first, self._kernel_expr, self._kernel_expr_cols)

    def _with_projected(self, batch: DeviceRecordBatch) -> DeviceRecordBatch:
        exprs = list(self._projected)
        used = {}
        for e in exprs:
            for c in ops.columns_of(e):
                used[c] = batch.columns[c]
        outs = ops.project_many(exprs, used, length=batch.num_rows)
        cols = dict(batch.columns)
        for e, o in zip(exprs, outs):
            cols[self._projected[e]] = o
        return DeviceRecordBatch(cols, batch.num_rows)

    def next(self):
        for batch in self._parent_operator.next():
            if self._agg is None:
                self._init(batch)
            if self._projected:
                batch = self._with_projected(batch)
            keys = [batch.column(c) for c in self._group_by]
            inputs = []
            for f in self._funcs:
                if f.expr is not None:
                    inputs.append(None if f.expr == self._kernel_expr else batch.column(self._projected[f.expr]))
                else:
                    inputs.append(batch.column(f.column) if f.column else None)
            pred = batch.column(self._fused_pred[0]) if self._fused_pred else None
            ecols = [batch.column(c) for c in self._kernel_expr_cols] if self._kernel_expr is not None else None
            self._agg.next(keys, inputs, pred=pred, nrows=batch.num_rows, expr_cols=ecols)
        if self._agg is not None:                                   # aggregate.py:121-122
            res = self._agg.result_arrays([self._group_by.index(c) for c in self._agg_cols], self._agg_cols,
                                          [f.out_name for f in self._funcs])
            self._agg.close()
            self._agg = None
            out = DeviceRecordBatch.from_arrow(res)
            for c in self._agg_cols:                       # dictionary-encoded keys (strings, bools ...): codes -> values later
                if self._key_dicts.get(c) is not None:
                    out.columns[c].dictionary = self._key_dicts[c]
            yield out
