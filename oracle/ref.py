"""ctypes wrapper around oracle/_ref/libvinum_ref.so — the REAL reference operators.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Only tests/, bench.py's
``cpu_baseline`` leg and tests/golden/gen_golden.py may import this module.

The class surface mirrors the reference's pybind11 module
(/root/reference/vinum/core/vinum_lib.cpp:20-165): ``next(batch)`` then one
``result()`` / ``sorted()``.
"""
import ctypes
import os

import pyarrow as pa

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libvinum_ref.so")

COUNT_STAR, COUNT, MIN, MAX, SUM, AVG = range(6)
ONE_GROUP, SINGLE, MULTI = range(3)
ASC, DESC = 0, 1

_lib = None


def available() -> bool:
    return os.path.exists(_LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        import pyarrow.compute  # noqa: F401  (loads libarrow_compute, registers kernels)
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.ref_last_error.restype = ctypes.c_char_p
        _lib.ref_agg_create.restype = ctypes.c_void_p
        _lib.ref_sort_create.restype = ctypes.c_void_p
        for name in ("ref_agg_next", "ref_agg_result", "ref_sort_next", "ref_sort_sorted"):
            getattr(_lib, name).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            getattr(_lib, name).restype = ctypes.c_int
        _lib.ref_agg_destroy.argtypes = [ctypes.c_void_p]
        _lib.ref_sort_destroy.argtypes = [ctypes.c_void_p]
    return _lib


def _cstrs(items):
    arr = (ctypes.c_char_p * max(len(items), 1))()
    for i, s in enumerate(items):
        arr[i] = s.encode()
    return arr


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


def _err():
    return RuntimeError(lib().ref_last_error().decode())


class RefAggregate:
    def __init__(self, kind, groupby_cols, agg_cols, funcs):
        """funcs: list of (func_id, in_col, out_col)."""
        L = lib()
        ftypes = (ctypes.c_int * max(len(funcs), 1))(*[f[0] for f in funcs])
        self._h = L.ref_agg_create(
            kind, len(groupby_cols), _cstrs(groupby_cols), len(agg_cols), _cstrs(agg_cols),
            len(funcs), ftypes, _cstrs([f[1] for f in funcs]), _cstrs([f[2] for f in funcs]))
        if not self._h:
            raise _err()

    def next(self, batch: pa.RecordBatch):
        c = _CStructs()
        batch._export_to_c(c.arr_ptr, c.sch_ptr)
        if lib().ref_agg_next(self._h, c.arr_ptr, c.sch_ptr):
            raise _err()

    def result(self) -> pa.RecordBatch:
        c = _CStructs()
        if lib().ref_agg_result(self._h, c.arr_ptr, c.sch_ptr):
            raise _err()
        return pa.RecordBatch._import_from_c(c.arr_ptr, c.sch_ptr)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ref_agg_destroy(self._h)
            self._h = None


class RefSort:
    def __init__(self, cols, orders):
        L = lib()
        ords = (ctypes.c_int * max(len(orders), 1))(*orders)
        self._h = L.ref_sort_create(len(cols), _cstrs(cols), ords)
        if not self._h:
            raise _err()

    def next(self, batch: pa.RecordBatch):
        c = _CStructs()
        batch._export_to_c(c.arr_ptr, c.sch_ptr)
        if lib().ref_sort_next(self._h, c.arr_ptr, c.sch_ptr):
            raise _err()

    def sorted(self) -> pa.RecordBatch:
        c = _CStructs()
        if lib().ref_sort_sorted(self._h, c.arr_ptr, c.sch_ptr):
            raise _err()
        return pa.RecordBatch._import_from_c(c.arr_ptr, c.sch_ptr)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ref_sort_destroy(self._h)
            self._h = None
