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

GenericHashAggregate (string / bool / decimal keys) is outside the GPU scope: constructing it
raises, it never silently falls back to a CPU implementation.
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

    def next(self, batch: pa.RecordBatch) -> None:
        c = _CStructs()
        batch._export_to_c(c.arr_ptr, c.sch_ptr)
        if L.lib().vnm_agg_op_next(self._h, c.arr_ptr, c.sch_ptr) != 0:
            raise RuntimeError(L.last_error())

    def result(self) -> pa.RecordBatch:
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


class GenericHashAggregate:
    def __init__(self, *args, **kwargs):
        raise RuntimeError("GenericHashAggregate (string/bool/decimal group keys) is not implemented on the MI355X path; "
                           "there is deliberately no CPU fallback (SURVEY.md §2 #1, §8f #3)")


class Sort:
    def __init__(self, sort_cols, sort_order):
        sort_cols, sort_order = list(sort_cols), [int(o) for o in sort_order]
        ords = (ctypes.c_int * max(len(sort_order), 1))(*sort_order)
        self._h = L.lib().vnm_sort_op_create(len(sort_cols), _cstrs(sort_cols), ords)
        if not self._h:
            raise RuntimeError(L.last_error())

    def next(self, batch: pa.RecordBatch) -> None:
        c = _CStructs()
        batch._export_to_c(c.arr_ptr, c.sch_ptr)
        if L.lib().vnm_sort_op_next(self._h, c.arr_ptr, c.sch_ptr) != 0:
            raise RuntimeError(L.last_error())

    def sorted(self, limit: int = 0) -> pa.RecordBatch:
        """limit (extension, default 0 = everything): only the first `limit` rows are needed (LIMIT pushed
        into the sort; identical rows to sorting everything and slicing)."""
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
