"""Drop-in mirror of the reference's pybind11 module ``vinum_lib`` (vinum/core/vinum_lib.cpp:20-165), backed by
libvinum_hip.so through the Arrow C Data Interface.

Same names, constructor arguments, methods and error behaviour as the reference:

    AggFuncType.{COUNT_STAR, COUNT, MIN, MAX, SUM, AVG}     (+ exported values)           :25-32
    SortOrder.{ASC, DESC}                                                                   :34-37
    AggFuncDef(func, column_name, out_col_name)                                             :39-51
    SingleNumericalHashAggregate / MultiNumericalHashAggregate (groupby_cols, agg_cols, agg_funcs)  :54-90
    OneGroupAggregate(agg_funcs)                                                            :111-124
    Sort(sort_cols, sort_order)           .next(batch) / .sorted()                          :126-142
    TableBatchReader(table)               .next() / .set_batch_size(n)                      :144-165
    import_pyarrow() -> 0                                                                   :22-23

    GenericHashAggregate(groupby_cols, agg_cols, agg_funcs)    string / bool / decimal keys          :92-109
        -- dictionary-encoded on ingest, grouped on the GPU by their int32 codes, decoded in result().
"""
import ctypes
import enum

import pyarrow as pa

from . import _lib as L


class AggFuncType(enum.IntEnum):
    COUNT_STAR = L.COUNT_STAR
    COUNT = L.COUNT
    MIN = L.MIN
    MAX = L.MAX
    SUM = L.SUM
    AVG = L.AVG


class SortOrder(enum.IntEnum):
    ASC = L.ASC
    DESC = L.DESC


# py::enum_<>::export_values()
COUNT_STAR, COUNT, MIN, MAX, SUM, AVG = (AggFuncType.COUNT_STAR, AggFuncType.COUNT, AggFuncType.MIN, AggFuncType.MAX,
                                         AggFuncType.SUM, AggFuncType.AVG)
ASC, DESC = SortOrder.ASC, SortOrder.DESC


def import_pyarrow() -> int:
    """0 on success (vinum/__init__.py:22-23 raises on non-zero).  Here: loads and initialises the HIP library."""
    L.lib()
    return 0


class AggFuncDef:
    def __init__(self, func, column_name: str, out_col_name: str):
        self.func = AggFuncType(int(func))
        self._column_name = column_name
        self._out_col_name = out_col_name

    @property
    def column_name(self):
        return self._column_name

    @property
    def out_col_name(self):
        return self._out_col_name

    def __repr__(self):
        return f"<AggFuncDef col_name: {self._column_name}, out_col_name: {self._out_col_name}>"


class _CStructs:
    """Scratch memory for one ArrowArray (80 B) + ArrowSchema (72 B)."""

    def __init__(self):
        self.arr = ctypes.create_string_buffer(80)
        self.sch = ctypes.create_string_buffer(72)

    @property
    def arr_ptr(self):
        return ctypes.addressof(self.arr)

    @property
    def sch_ptr(self):
        return ctypes.addressof(self.sch)


def _cstrs(items):
    arr = (ctypes.c_char_p * max(len(items), 1))()
    for i, s in enumerate(items):
        arr[i] = s.encode()
    return arr


class _HashAggregateBase:
    _KIND = None

    def __init__(self, groupby_cols, agg_cols, agg_funcs):
        lib = L.lib()
        groupby_cols, agg_cols, agg_funcs = list(groupby_cols), list(agg_cols), list(agg_funcs)
        ftypes = (ctypes.c_int * max(len(agg_funcs), 1))(*[int(f.func) for f in agg_funcs])
        self._h = lib.vnm_agg_op_create(self._KIND, len(groupby_cols), _cstrs(groupby_cols), len(agg_cols),
                                        _cstrs(agg_cols), len(agg_funcs), ftypes,
                                        _cstrs([f.column_name for f in agg_funcs]),
                                        _cstrs([f.out_col_name for f in agg_funcs]))
        if not self._h:
            raise RuntimeError(L.last_error())

    # The reference streams 10 000-row batches by default (vinum/__init__.py:52).  Handing each of them to the library costs ~50 us
    # of Python + Arrow C export per batch -- 20x what its rows cost on the device -- so batches below 2^20 rows are kept here
    # (a list append) and cross the boundary joined into one batch once 2^22 rows are waiting, or at result().  (The C entry
    # point coalesces small batches as well, for callers that are not this class.)
    _SMALL_ROWS = 1 << 20
    _FLUSH_ROWS = 1 << 22

    def _send(self, batch: pa.RecordBatch) -> None:
        c = _CStructs()
        batch._export_to_c(c.arr_ptr, c.sch_ptr)
        if L.lib().vnm_agg_op_next(self._h, c.arr_ptr, c.sch_ptr) != 0:
            raise RuntimeError(L.last_error())

    def _flush(self) -> None:
        pending, self._pending, self._pending_rows = getattr(self, "_pending", None), [], 0
        if pending:
            joined = pa.Table.from_batches(pending).combine_chunks()
            for b in joined.to_batches():
                self._send(b)

    def next(self, batch: pa.RecordBatch) -> None:
        if not hasattr(self, "_pending"):
            # the FIRST batch goes straight through: it fixes the schema, and an unknown column or an unsupported type must raise
            # from this call as it does in the reference (base_aggregate.cpp:91-131)
            self._pending, self._pending_rows = [], 0
            self._send(batch)
            return
        if batch.num_rows >= self._SMALL_ROWS:
            self._flush()
            self._send(batch)
            return
        if self._pending and batch.schema != self._pending[0].schema:
            self._flush()
        self._pending.append(batch)
        self._pending_rows += batch.num_rows
        if self._pending_rows >= self._FLUSH_ROWS:
            self._flush()

    def result(self) -> pa.RecordBatch:
        self._flush()
        c = _CStructs()
        if L.lib().vnm_agg_op_result(self._h, c.arr_ptr, c.sch_ptr) != 0:
            raise RuntimeError(L.last_error())
        return pa.RecordBatch._import_from_c(c.arr_ptr, c.sch_ptr)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.lib().vnm_agg_op_destroy(h)
            except Exception:
                pass


class SingleNumericalHashAggregate(_HashAggregateBase):
    _KIND = L.SINGLE_NUMERICAL


class MultiNumericalHashAggregate(_HashAggregateBase):
    _KIND = L.MULTI_NUMERICAL


class OneGroupAggregate(_HashAggregateBase):
    _KIND = L.ONE_GROUP

    def __init__(self, agg_funcs):
        super().__init__([], [], agg_funcs)


class KeyDictionary:
    """Running dictionary of one non-numeric group-by column: value -> int32 code, in order of first appearance.

    The reference keys its generic map on arrow::Scalar vectors (generic_hash_aggregate.h:10-45): hash + Equals per row.
    Here every batch is dictionary-encoded once on ingest (Arrow's C++ `dictionary_encode`, one pass), the batch-local
    codes are mapped onto the running dictionary, and the GPU groups by 4-byte codes with the numeric machinery
    (NULL stays NULL: its own group, as in the reference where a NULL scalar equals a NULL scalar)."""

    def __init__(self, arrow_type: pa.DataType):
        self.type = arrow_type
        self.values = pa.array([], type=arrow_type)

    def encode(self, column) -> pa.Array:
        import numpy as np
        import pyarrow.compute as pc
        if isinstance(column, pa.ChunkedArray):
            column = column.combine_chunks()
        enc = column.dictionary_encode()
        local = enc.dictionary
        pos = pc.index_in(local, value_set=self.values) if len(self.values) else pa.nulls(len(local), pa.int32())
        known = pos.is_valid().to_numpy(zero_copy_only=False)
        mapping = pos.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int32)
        n_new = int((~known).sum())
        if n_new:
            mapping[~known] = len(self.values) + np.arange(n_new, dtype=np.int32)
            fresh = local.filter(pa.array(~known))
            self.values = pa.concat_arrays([self.values, fresh]) if len(self.values) else fresh
        idx = enc.indices
        codes = mapping[idx.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int64)] if len(local) else np.zeros(len(idx), np.int32)
        mask = ~idx.is_valid().to_numpy(zero_copy_only=False) if idx.null_count else None
        return pa.array(codes, type=pa.int32(), mask=mask)

    def decode(self, codes: pa.Array) -> pa.Array:
        return self.values.take(codes)


def _is_numeric(t: pa.DataType) -> bool:        # vinum/core/aggregate.py:63-66
    return pa.types.is_integer(t) or pa.types.is_floating(t) or pa.types.is_temporal(t)


class GenericHashAggregate:
    """GROUP BY over keys of ANY Arrow type -- strings, bools, decimals, mixed with numeric keys
    (vinum/core/vinum_lib.cpp:92-109, vinum_cpp/src/operators/aggregate/generic_hash_aggregate.{h,cpp}).

    Non-numeric key columns are dictionary-encoded on ingest (KeyDictionary) and the query runs through the numeric GPU
    operators (single key: the whole single-key machinery incl. the partitioned paths; several keys: packed composite
    keys); result() maps the codes back, so key columns keep their type.  Aggregate INPUT columns must be numeric --
    COUNT(col) of a non-numeric column counts through an int8 stand-in with the same validity; MIN / MAX of strings
    (agg_funcs.h:219-261) are not available on the GPU path and raise."""

    def __init__(self, groupby_cols, agg_cols, agg_funcs):
        self._groupby, self._agg_cols, self._funcs = list(groupby_cols), list(agg_cols), list(agg_funcs)
        self._inner = None
        self._dicts = {}
        self._stand_in = set()

    def _init(self, batch: pa.RecordBatch):
        schema = batch.schema
        for c in self._groupby:
            if c not in schema.names:
                raise RuntimeError(f"Column not found: {c}")       # base_aggregate.cpp:121-131
            t = schema.field(c).type
            if not _is_numeric(t):
                self._dicts[c] = KeyDictionary(t)
        for f in self._funcs:
            if f.column_name and f.column_name in schema.names and not _is_numeric(schema.field(f.column_name).type):
                if f.func != AggFuncType.COUNT:
                    raise RuntimeError({AggFuncType.MIN: "Column data type is not supported by min()/max().",
                                        AggFuncType.MAX: "Column data type is not supported by min()/max().",
                                        AggFuncType.SUM: "Column data type is not supported by sum().",
                                        AggFuncType.AVG: "Column data type is not supported by avg()."}[f.func]
                                       + " (non-numeric aggregate inputs are CPU-only in the reference; not on the GPU path)")
                self._stand_in.add(f.column_name)
        cls = SingleNumericalHashAggregate if len(self._groupby) == 1 else MultiNumericalHashAggregate
        self._inner = cls(self._groupby, self._agg_cols, self._funcs)

    def next(self, batch: pa.RecordBatch) -> None:
        # the first batch fixes the schema (and raises what the reference raises); later small batches are encoded together:
        # dictionary-encoding 10 000 rows at a time costs more in Python than in work
        if self._inner is None:
            self._init(batch)
            self._pending, self._pending_rows = [], 0
            self._encode_and_send(batch)
            return
        if batch.num_rows >= (1 << 20):
            self._flush()
            self._encode_and_send(batch)
            return
        if self._pending and batch.schema != self._pending[0].schema:
            self._flush()
        self._pending.append(batch)
        self._pending_rows += batch.num_rows
        if self._pending_rows >= (1 << 22):
            self._flush()

    def _flush(self) -> None:
        pending, self._pending, self._pending_rows = self._pending, [], 0
        if pending:
            for b in pa.Table.from_batches(pending).combine_chunks().to_batches():
                self._encode_and_send(b)

    def _encode_and_send(self, batch: pa.RecordBatch) -> None:
        import numpy as np
        arrays, names = [], []
        for i, name in enumerate(batch.schema.names):
            col = batch.column(i)
            if name in self._dicts:
                col = self._dicts[name].encode(col)
            elif name in self._stand_in:
                col = pa.array(np.zeros(len(col), np.int8), mask=(~col.is_valid().to_numpy(zero_copy_only=False)) if col.null_count else None)
            elif not _is_numeric(col.type):
                continue                                     # neither a key nor an input: never staged
            arrays.append(col)
            names.append(name)
        self._inner.next(pa.RecordBatch.from_arrays(arrays, names=names))

    def result(self) -> pa.RecordBatch:
        if self._inner is None:
            raise RuntimeError("GenericHashAggregate.result() before any batch")
        self._flush()
        res = self._inner.result()
        arrays = [self._dicts[n].decode(res.column(i)) if n in self._dicts else res.column(i) for i, n in enumerate(res.schema.names)]
        return pa.RecordBatch.from_arrays(arrays, names=res.schema.names)


class Sort:
    def __init__(self, sort_cols, sort_order):
        sort_cols, sort_order = list(sort_cols), [int(o) for o in sort_order]
        ords = (ctypes.c_int * max(len(sort_order), 1))(*sort_order)
        self._h = L.lib().vnm_sort_op_create(len(sort_cols), _cstrs(sort_cols), ords)
        if not self._h:
            raise RuntimeError(L.last_error())

    def next(self, batch: pa.RecordBatch) -> None:
        # Sort::Next only retains the batch (sort.cpp:11-13); nothing can fail before sorted().  Small batches (the reference's
        # default is 10 000 rows) are kept here and cross the boundary joined, as in the aggregates.
        if not hasattr(self, "_pending"):
            self._pending, self._pending_rows = [], 0
        if self._pending and batch.schema != self._pending[0].schema:
            self._flush()
        self._pending.append(batch)
        self._pending_rows += batch.num_rows
        if self._pending_rows >= (1 << 24):
            self._flush()

    def _flush(self) -> None:
        pending, self._pending, self._pending_rows = getattr(self, "_pending", None), [], 0
        if not pending:
            return
        joined = pa.Table.from_batches(pending).combine_chunks() if len(pending) > 1 else pa.Table.from_batches(pending)
        for b in (joined.to_batches() or pending[:1]):
            c = _CStructs()
            b._export_to_c(c.arr_ptr, c.sch_ptr)
            if L.lib().vnm_sort_op_next(self._h, c.arr_ptr, c.sch_ptr) != 0:
                raise RuntimeError(L.last_error())

    def sorted(self, limit: int = 0) -> pa.RecordBatch:
        """limit (extension, default 0 = everything): only the first `limit` rows are needed (LIMIT pushed
        into the sort; identical rows to sorting everything and slicing)."""
        self._flush()
        c = _CStructs()
        if L.lib().vnm_sort_op_sorted(self._h, int(limit), c.arr_ptr, c.sch_ptr) != 0:
            raise RuntimeError(L.last_error())
        return pa.RecordBatch._import_from_c(c.arr_ptr, c.sch_ptr)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.lib().vnm_sort_op_destroy(h)
            except Exception:
                pass


class TableBatchReader:
    """arrow::TableBatchReader wrapper (vinum_cpp/src/operators/table_batch_reader.cpp:5-16): zero-copy slices of
    `batch_size` rows, never crossing a chunk boundary; None at the end.  Pure host bookkeeping -- the H2D staging
    happens when an operator consumes the batch."""

    def __init__(self, table: pa.Table):
        self._table = table
        self._batch_size = None
        self._iter = None

    def set_batch_size(self, batch_size: int) -> None:
        self._batch_size = int(batch_size)
        self._iter = None

    def next(self):
        if self._iter is None:
            self._iter = iter(self._table.to_batches(max_chunksize=self._batch_size))
        return next(self._iter, None)
