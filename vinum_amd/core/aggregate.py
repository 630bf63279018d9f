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
    def __init__(self, func: str, column: Optional[str] = None, out_name: Optional[str] = None):
        self.func = func.upper()
        if self.func == "COUNT" and not column:
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
        for f in self._funcs:
            if f.column:
                spec.append((_FUNCS[f.func], names.index(f.column), batch.column(f.column).arrow_type))
            else:
                spec.append((_FUNCS[f.func], None, None))
        self._agg = ops.DeviceAggregate(kind, key_types, spec, expected_groups=self._expected_groups)
        if self._fused_pred:
            self._agg.set_predicate(self._fused_pred[1], self._fused_pred[2])

    def next(self):
        for batch in self._parent_operator.next():
            if self._agg is None:
                self._init(batch)
            keys = [batch.column(c) for c in self._group_by]
            inputs = [batch.column(f.column) if f.column else None for f in self._funcs]
            pred = batch.column(self._fused_pred[0]) if self._fused_pred else None
            self._agg.next(keys, inputs, pred=pred, nrows=batch.num_rows)
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
