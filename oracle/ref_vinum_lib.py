"""`vinum_lib` as the reference's Python layer imports it (vinum/__init__.py:21, core/aggregate.py:5, core/algebra.py:14),
backed by the REAL reference operators through oracle/_ref (oracle/ref.py) instead of the pybind11 module
(vinum/core/vinum_lib.cpp:20-165), which would need a 45 s pybind build for the same C++ code.

TEST INFRASTRUCTURE ONLY (build container): tests/golden/gen_golden_planner.py puts this module into
sys.modules['vinum_lib'] so that the reference's unchanged planner / executor produce whole-query fixtures.
GenericHashAggregate is absent: generic_hash_aggregate.h does not compile against Arrow 25."""
import enum

import pyarrow as pa

from oracle import ref


def import_pyarrow():
    return 0


class AggFuncType(enum.IntEnum):
    COUNT_STAR = 0
    COUNT = 1
    MIN = 2
    MAX = 3
    SUM = 4
    AVG = 5


class SortOrder(enum.IntEnum):
    ASC = 0
    DESC = 1


class AggFuncDef:
    def __init__(self, func, column_name, out_col_name):
        self.func, self.column_name, self.out_col_name = func, column_name, out_col_name


def _funcs(defs):
    return [(int(d.func), d.column_name, d.out_col_name) for d in defs]


class _Agg:
    KIND = None

    def __init__(self, groupby_cols, agg_cols, agg_funcs):
        self._a = ref.RefAggregate(self.KIND, list(groupby_cols), list(agg_cols), _funcs(agg_funcs))

    def next(self, batch):
        self._a.next(batch)

    def result(self):
        return self._a.result()


class SingleNumericalHashAggregate(_Agg):
    KIND = ref.SINGLE


class MultiNumericalHashAggregate(_Agg):
    KIND = ref.MULTI


class OneGroupAggregate(_Agg):
    KIND = ref.ONE_GROUP

    def __init__(self, agg_funcs):
        self._a = ref.RefAggregate(ref.ONE_GROUP, [], [], _funcs(agg_funcs))


class GenericHashAggregate:
    def __init__(self, *a, **k):
        raise RuntimeError("GenericHashAggregate is not buildable against Arrow 25 (oracle/ref_build/Makefile)")


class Sort:
    def __init__(self, sort_cols, sort_order):
        self._s = ref.RefSort(list(sort_cols), [int(o) for o in sort_order])

    def next(self, batch):
        self._s.next(batch)

    def sorted(self):
        return self._s.sorted()


class TableBatchReader:
    """table_batch_reader.cpp:5-16 is arrow::TableBatchReader with a chunk size: pyarrow exposes the same class."""

    def __init__(self, table: pa.Table):
        self._table = table
        self._size = 1 << 20
        self._it = None

    def set_batch_size(self, n):
        self._size = int(n)

    def next(self):
        if self._it is None:
            self._it = iter(self._table.to_batches(max_chunksize=self._size))
        return next(self._it, None)
