"""GPU counterparts of the reference's Python physical operators (vinum/core/algebra.py)."""
from typing import Iterable, List, Optional, Sequence, Tuple

import pyarrow as pa

from .. import get_batch_size, ops
from .. import _lib as L
from ..device import DeviceColumn, is_supported
from .base import DeviceRecordBatch, Operator


class TableReaderOperator(Operator):
    """algebra.py:250-265 + table_batch_reader.cpp: slices the table into batches of vinum_amd.get_batch_size() rows
    (2^24 by default -- the reference's 10 000 would mean 1e5 launches per 1e9 rows) and stages them into HBM.
    `columns` prunes what is staged (the planner's pruning projection, planner.py:367-371)."""

    def __init__(self, table: pa.Table, columns: Optional[Sequence[str]] = None):
        super().__init__(None)
        self._table = table.select(list(columns)) if columns is not None else table
        self._dicts = {}     # running dictionaries of the non-numeric columns (shared by every batch of this scan)

    def next(self):
        for b in self._table.to_batches(max_chunksize=get_batch_size()):
            yield DeviceRecordBatch.from_arrow(b, self._dicts)


class FileReaderOperator(Operator):
    """algebra.py:268-279: pulls from a pyarrow streaming reader (pyarrow.csv.open_csv, io/arrow.py:58-61) until
    StopIteration.  Only GPU-representable columns are staged."""

    def __init__(self, reader, columns: Optional[Sequence[str]] = None):
        super().__init__(None)
        self._reader, self._columns = reader, columns
        self._dicts = {}

    def next(self):
        while True:
            if hasattr(self._reader, "read_next_device_batch"):     # vinum_amd.io.GpuCsvReader: columns are born in HBM
                try:
                    yield self._reader.read_next_device_batch()
                except StopIteration:
                    break
                continue
            try:
                batch = self._reader.read_next_batch()
            except StopIteration:
                break
            names = self._columns if self._columns is not None else [f.name for f in batch.schema if is_supported(f.type)]
            yield DeviceRecordBatch.from_arrow(batch.select(list(names)), self._dicts)


class FilterOperator(Operator):
    """algebra.py:108-123: `WHERE column <op> literal`; every column of the batch is compacted (no selection
    vectors in the reference either).  predicate = (column, op, literal)."""

    def __init__(self, predicate: Tuple[str, str, object], parent_operator: Operator):
        super().__init__(parent_operator)
        self.predicate = predicate

    @staticmethod
    def is_simple(pred) -> bool:
        """`column <op> literal`: the shape that runs fused (and can be folded into an aggregate scan)."""
        return (isinstance(pred, tuple) and len(pred) == 3 and isinstance(pred[0], str) and pred[1] in ops.CMP_OPS
                and isinstance(pred[2], (int, float)) and not isinstance(pred[2], bool))

    _FLIP = {"lt": "gt", "le": "ge", "gt": "lt", "ge": "le", "eq": "eq", "ne": "ne"}
    _SYM = {"==": "eq", "!=": "ne", ">": "gt", ">=": "ge", "<": "lt", "<=": "le"}

    @classmethod
    def lower_dictionary_predicates(cls, pred, batch: DeviceRecordBatch):
        """Comparisons of a dictionary-coded column (strings, binaries: int32 codes in HBM) with a LITERAL -- `city = 'Berlin'`,
        `name < 'M'`, `city IN ('a', 'b')`; literals are ("lit", value) nodes, a bare str is a column name -- become numeric
        comparisons: = / != / IN on the literal's CODE (a value the dictionary does not hold matches no row), < <= > >= on the
        column's order-preserving RANKS (KeyDictionary.rank_column: byte-wise order, as Arrow / NumPy compare such values) against
        the literal's position among the dictionary's values.  NULL rows behave as in every other predicate (compare False,
        `!=` True: vinum/arrow/record_batch.py:112-118).  Returns (predicate, extra columns the predicate reads)."""
        extra = {}

        def is_dict(x):
            return isinstance(x, str) and x in batch.columns and batch.columns[x].dictionary is not None

        def is_lit(x):
            return isinstance(x, tuple) and len(x) == 2 and x[0] == "lit" and isinstance(x[1], (str, bytes))

        def walk(e):
            if not isinstance(e, tuple) or not e:
                return e
            op = e[0]
            if op in cls._FLIP and len(e) == 3:
                a, b = e[1], e[2]
                if is_lit(a) and is_dict(b):
                    a, b, op = b, a, cls._FLIP[op]
                if is_dict(a) and is_lit(b):
                    d = batch.columns[a].dictionary
                    if op in ("eq", "ne"):
                        c = d.code_of(b[1])
                        return (op, a, -2 if c is None else c)            # (codes are >= 0)
                    name = f"__rank_{a}"
                    if name not in extra:
                        extra[name] = d.rank_column(batch.columns[a])
                    less, present = d.rank_bounds(b[1])
                    return {"lt": ("lt", name, less), "le": ("lt", name, less + present),
                            "gt": ("ge", name, less + present), "ge": ("ge", name, less)}[op]
            if op in ("in", "not_in") and is_dict(e[1]) and all(isinstance(v, (str, bytes)) or is_lit(v) for v in e[2]):
                d = batch.columns[e[1]].dictionary
                codes = [d.code_of(v[1] if is_lit(v) else v) for v in e[2]]
                return (op, e[1], tuple(c for c in codes if c is not None) or (-2,))
            if op in ("in", "not_in"):
                return (op, walk(e[1]), e[2])
            return tuple([op] + [walk(x) for x in e[1:]])
        return walk(pred), extra

    def _kernel(self, batch: DeviceRecordBatch) -> DeviceRecordBatch:
        names = batch.column_names
        cols = [batch.columns[n] for n in names]
        pred = self.predicate
        if self.is_simple(pred) or (isinstance(pred, tuple) and len(pred) == 3 and pred[1] in self._SYM and isinstance(pred[2], (str, bytes))):
            col, op, lit = pred
            if isinstance(lit, (str, bytes)):       # (column, "==", "Berlin") over a dictionary-coded column
                pred = (self._SYM[op], col, ("lit", lit))
            else:
                outs, k = ops.filter_cmp(batch.column(col), op, lit, cols)
                return DeviceRecordBatch(dict(zip(names, outs)), k)
        pred, extra = self.lower_dictionary_predicates(pred, batch)
        # general boolean expression tree: one fused mask kernel, then one compaction pass
        columns = dict(batch.columns)
        columns.update(extra)
        mask = ops.predicate_mask(pred, columns, batch.num_rows)
        outs, k = ops.filter_mask(mask, None, batch.num_rows, cols)
        return DeviceRecordBatch(dict(zip(names, outs)), k)


class ProjectOperator(Operator):
    """algebra.py:28-105: expressions (prefix tuples, see ops.compile_expr) or plain column names -> output columns.
    keep_input_table appends instead of replacing (:56-62)."""

    def __init__(self, arguments: Sequence, parent_operator: Operator, col_names: Optional[Sequence[str]] = None,
                 keep_input_table: bool = False):
        super().__init__(parent_operator)
        self._arguments = list(arguments)
        self._col_names = list(col_names) if col_names is not None else [a if isinstance(a, str) else f"expr_{i}"
                                                                          for i, a in enumerate(arguments)]
        self._keep = keep_input_table

    def _kernel(self, batch: DeviceRecordBatch) -> DeviceRecordBatch:
        out = dict(batch.columns) if self._keep else {}
        n = batch.num_rows
        # every computed expression of the SELECT list goes into one fused kernel
        exprs, used = [], {}
        for name, arg in zip(self._col_names, self._arguments):
            if not isinstance(arg, str):
                exprs.append((name, arg))
                for c in _columns_of(arg):
                    used[c] = batch.columns[c]
        computed = {}
        if exprs:
            # column-free expressions repeat their scalar once per row (:77-87)
            res = ops.project_many([e for _, e in exprs], used, length=n if used else max(n, 1))
            computed = {name: col for (name, _), col in zip(exprs, res)}
        for name, arg in zip(self._col_names, self._arguments):
            out[name] = batch.column(arg) if isinstance(arg, str) else computed[name]
        return DeviceRecordBatch(out, n if out and any(not isinstance(a, (int, float)) for a in self._arguments) else n)


def _columns_of(expr) -> List[str]:
    return ops.columns_of(expr)


class SortOperator(Operator):
    """algebra.py:126-201 + sort.cpp: buffers every batch, one stable multi-key sort, all columns taken.
    limit > 0 pushes LIMIT into the sort (same rows as sorting everything, then SliceOperator)."""

    def __init__(self, columns: Sequence[str], sort_order: Sequence[int], parent_operator: Operator, limit: int = 0):
        super().__init__(parent_operator)
        self._cols, self._orders, self._limit = list(columns), [int(o) for o in sort_order], int(limit)

    def next(self):
        batches = [b.to_arrow() for b in self._parent_operator.next()]
        if not batches:
            return
        table = pa.Table.from_batches(batches).combine_chunks()       # Table::FromRecordBatches (sort.cpp:16)
        for f in table.schema:
            if f.name in self._cols and pa.types.is_boolean(f.type):   # algebra.py:191-201
                raise RuntimeError("Sorting by boolean column is not supported yet. "
                                   "Please use float(bool_column) as a workaround.")
        dicts = {}
        dev = DeviceRecordBatch.from_arrow(table.to_batches()[0] if table.num_rows else pa.RecordBatch.from_arrays(
            [pa.array([], f.type) for f in table.schema], names=table.schema.names), dicts).columns
        n = table.num_rows
        k = self._limit if 0 < self._limit < n else 0
        m = k if k else n
        # a string / binary / decimal ... sort key is dictionary-coded in HBM: its order-preserving ranks are the key (NULL stays
        # NULL, equal values share a rank -- the stable sort keeps their row order like SortIndices, sort.cpp:22-37)
        keys = [dev[c] if dev[c].dictionary is None else dev[c].dictionary.rank_column(dev[c]) for c in self._cols]
        sorted_key = None
        if k:
            idx = ops.sort_indices(keys, self._orders, limit=k)
        else:   # a full sort hands its first key back sorted: one gather less
            idx, sorted_key = ops.sort_indices_keyed(keys, self._orders)
        if dev[self._cols[0]].dictionary is not None:
            sorted_key = None          # (the sorted ranks are not the column)
        yield DeviceRecordBatch({name: sorted_key if (sorted_key is not None and name == self._cols[0]) else ops.take(col, idx, m)
                                 for name, col in dev.items()}, m)


class SliceOperator(Operator):
    """LIMIT / OFFSET over the batch stream (algebra.py:204-247).  Rows are counted across batches; a batch that is only
    partly wanted is narrowed to a VIEW of the same HBM buffers (Arrow offset + length, as pa.RecordBatch.slice does on
    the host) -- nothing is copied and nothing leaves the device."""

    def __init__(self, limit: int, offset: int, parent_operator: Operator):
        super().__init__(parent_operator)
        self._limit, self._offset = int(limit), int(offset)

    def next(self):
        skip, want = self._offset, self._limit
        for batch in self._parent_operator.next():
            if want <= 0:
                break
            n = batch.num_rows
            if skip >= n:
                skip -= n
                continue
            take = min(want, n - skip)
            yield batch if (skip == 0 and take == n) else batch.slice(skip, take)
            want -= take
            skip = 0


class MaterializeTableOperator(Operator):
    """algebra.py:290-295: pulls everything, returns one pyarrow Table (the only D2H of the pipeline)."""

    def next(self):
        batches = [b.to_arrow() for b in self._parent_operator.next()]
        yield pa.Table.from_batches(batches) if batches else pa.table({})
