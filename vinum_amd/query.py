"""The build's counterpart of the reference's planner + executor for the hot-path query shapes
(vinum/planner/planner.py:330-507, vinum/executor/executor.py:30-31): builds the operator chain in the
planner's order and pulls it once.

    reader -> [Filter] -> [Project pre-agg] -> [Aggregate] -> [Sort] -> Project(final) -> [Slice] -> Materialize

SQL parsing (pglast) is out of scope; callers pass the query pieces directly.
"""
from typing import Optional, Sequence, Tuple

import pyarrow as pa

from .core import (AggregateFunction, AggregateOperator, FileReaderOperator, FilterOperator, MaterializeTableOperator,
                   ProjectOperator, SliceOperator, SortOperator, TableReaderOperator)


def select(source, columns: Optional[Sequence] = None, where: Optional[Tuple[str, str, object]] = None,
           group_by: Sequence[str] = (), aggregates: Sequence[AggregateFunction] = (),
           order_by: Sequence[str] = (), sort_order: Sequence[int] = (), limit: Optional[int] = None, offset: int = 0,
           expected_groups: int = 0) -> pa.Table:
    """source: pyarrow.Table or a pyarrow streaming reader (stream_csv).  columns: output columns / expressions as
    (name, expr) pairs or plain names; with aggregates, the selected group-by columns."""
    needed = _needed_columns(columns, where, group_by, aggregates, order_by)
    if isinstance(source, pa.Table):
        op = TableReaderOperator(source, columns=needed)
    else:
        op = FileReaderOperator(source, columns=needed)
    if where is not None:
        op = FilterOperator(where, op)
    if aggregates or group_by:
        agg_cols = [c for c in (columns or group_by) if isinstance(c, str) and c in group_by]
        op = AggregateOperator(op, list(group_by), list(aggregates), agg_cols, expected_groups=expected_groups)
    if order_by:
        op = SortOperator(list(order_by), list(sort_order) or [0] * len(order_by), op, limit=(limit or 0) + offset if limit else 0)
    if columns and not (aggregates or group_by):
        names = [c if isinstance(c, str) else c[0] for c in columns]
        args = [c if isinstance(c, str) else c[1] for c in columns]
        op = ProjectOperator(args, op, col_names=names)
    if limit is not None:
        op = SliceOperator(limit, offset, op)
    return next(MaterializeTableOperator(op).next())


def _needed_columns(columns, where, group_by, aggregates, order_by):
    from .core.algebra import _columns_of
    need = []

    def add(c):
        if c and c not in need:
            need.append(c)
    for c in (columns or []):
        for x in ([c] if isinstance(c, str) else _columns_of(c[1])):
            add(x)
    if where:
        from .ops import columns_of
        from .core.algebra import FilterOperator
        for c in ([where[0]] if FilterOperator.is_simple(where) else columns_of(where)):
            add(c)
    for c in group_by:
        add(c)
    for f in aggregates:
        add(f.column)
    for c in order_by:
        add(c)
    return need or None
