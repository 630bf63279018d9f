"""pyarrow-facing wrapper of the CPU oracle (oracle/vinum_oracle.c).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import this; the product (vinum_amd/) never does.

The classes mirror the reference's pybind11 surface
(/root/reference/vinum/core/vinum_lib.cpp:54-142): ``next(batch)`` per batch and
one ``result()`` / ``sorted()``; column order and output types follow
base_aggregate.cpp:47-68 and agg_func_factory.cpp:13-329.
"""
import ctypes
import os
import subprocess

import numpy as np
import pyarrow as pa

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "vinum_oracle.c")
_SO = os.path.join(_HERE, "_build", "libvinum_oracle.so")

COUNT_STAR, COUNT, MIN, MAX, SUM, AVG = range(6)
ONE_GROUP, SINGLE, MULTI = range(3)
ASC, DESC = 0, 1
EQ, NE, GT, GE, LT, LE = range(6)
I8, I16, I32, I64, U8, U16, U32, U64, F32, F64 = range(10)
OUT_U64, OUT_I64, OUT_F64, OUT_F32, OUT_DEC128, OUT_I32 = range(6)
FLAG_SUM32 = 1

_NP = {I8: np.int8, I16: np.int16, I32: np.int32, I64: np.int64, U8: np.uint8, U16: np.uint16,
       U32: np.uint32, U64: np.uint64, F32: np.float32, F64: np.float64}


def build(force=False):
    """gcc -O2 -shared oracle/vinum_oracle.c -> oracle/_build/libvinum_oracle.so
    VNM_ORACLE_SANITIZE=1 (tests/test_oracle_golden.py::test_oracle_under_asan_ubsan): an -O1 -fsanitize=address,undefined
    build in its own file; the process must have libasan preloaded."""
    if os.environ.get("VNM_ORACLE_SANITIZE") == "1":
        so = os.path.join(_HERE, "_build", "libvinum_oracle_asan.so")
        if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(_SRC):
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu11", "-fPIC", "-shared", "-fsanitize=address,undefined",
                                   "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-o", so, _SRC, "-lm"])
        return so
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(_SRC):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-o", _SO, _SRC, "-lm"])
    return _SO


class _Col(ctypes.Structure):
    _fields_ = [("values", ctypes.c_void_p), ("validity", ctypes.c_void_p), ("offset", ctypes.c_int64),
                ("length", ctypes.c_int64), ("type", ctypes.c_int32), ("flags", ctypes.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.orc_agg_create.restype = ctypes.c_void_p
        L.orc_agg_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p]
        L.orc_agg_destroy.argtypes = [ctypes.c_void_p]
        L.orc_agg_next.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_agg_ngroups.restype = ctypes.c_int64
        L.orc_agg_ngroups.argtypes = [ctypes.c_void_p]
        L.orc_agg_keys.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_agg_func.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_cmp_mask.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int64,
                                   ctypes.c_int, ctypes.c_void_p]
        L.orc_filter_col.restype = ctypes.c_int64
        L.orc_filter_col.argtypes = [ctypes.c_void_p] * 5
        L.orc_sort_indices.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_void_p]
        L.orc_take_col.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                   ctypes.c_void_p]
        _lib = L
    return _lib


def physical_type(t: pa.DataType):
    """Arrow type -> (physical id, flags).  Temporal types map to their storage integers."""
    T = pa.types
    if T.is_int8(t): return I8, 0
    if T.is_int16(t): return I16, 0
    if T.is_int32(t) or T.is_date32(t): return I32, 0
    if T.is_time32(t): return I32, FLAG_SUM32
    if T.is_int64(t) or T.is_date64(t) or T.is_time64(t) or T.is_timestamp(t) or T.is_duration(t): return I64, 0
    if T.is_uint8(t): return U8, 0
    if T.is_uint16(t): return U16, 0
    if T.is_uint32(t): return U32, 0
    if T.is_uint64(t): return U64, 0
    if T.is_float32(t): return F32, 0
    if T.is_float64(t): return F64, 0
    raise TypeError(f"oracle: unsupported column type {t}")


def _col(arr: pa.Array, keep):
    """Zero-copy view of a primitive Arrow array."""
    if isinstance(arr, pa.ChunkedArray):
        arr = arr.combine_chunks()
    pt, fl = physical_type(arr.type)
    bufs = arr.buffers()
    keep.append(arr)
    c = _Col()
    c.values = bufs[1].address if bufs[1] is not None else None
    c.validity = bufs[0].address if (bufs[0] is not None and arr.null_count > 0) else None
    c.offset = arr.offset
    c.length = len(arr)
    c.type = pt
    c.flags = fl
    return c


def _arrow_from(values: np.ndarray, valid: np.ndarray, t: pa.DataType) -> pa.Array:
    mask = ~valid.astype(bool)
    if pa.types.is_decimal(t):
        import decimal
        ctx = decimal.Context(prec=60)
        out = []
        for i in range(len(valid)):
            if mask[i]:
                out.append(None)
            else:
                lo = int(values[i, 0]); hi = int(values[i, 1].astype(np.int64))
                out.append(ctx.create_decimal(hi * (1 << 64) + lo))
        return pa.array(out, type=t)
    storage = pa.array(values, mask=mask if mask.any() else None)
    if storage.type != t:
        storage = storage.view(t) if _same_width(storage.type, t) else storage.cast(t)
    return storage


def _same_width(a, b):
    try:
        return a.bit_width == b.bit_width
    except Exception:
        return False


def _minmax_out(values_u64: np.ndarray, valid, t: pa.DataType):
    pt, _ = physical_type(t)
    if pt in (F32, F64):
        v = values_u64.view(np.float64).astype(_NP[pt])
    elif pt in (U8, U16, U32, U64):
        v = values_u64.astype(_NP[pt])
    else:
        v = values_u64.view(np.int64).astype(_NP[pt])
    return _arrow_from(v, valid, t)


class OracleAggregate:
    """kind in {ONE_GROUP, SINGLE, MULTI}; funcs: list of (func_id, in_col, out_col)."""

    def __init__(self, kind, groupby_cols, agg_cols, funcs):
        self.kind, self.groupby_cols, self.agg_cols, self.funcs = kind, list(groupby_cols), list(agg_cols), list(funcs)
        self._h = None
        self._schema = None

    def _init(self, schema: pa.Schema):
        self._schema = schema
        for c in self.groupby_cols + self.agg_cols:
            if schema.get_field_index(c) < 0:
                raise RuntimeError("Column not found: " + c)  # base_aggregate.cpp:121-131
        kt = [physical_type(schema.field(c).type)[0] for c in self.groupby_cols]
        ft, it, fl = [], [], []
        for f, col, _ in self.funcs:
            ft.append(f)
            if col:
                p, g = physical_type(schema.field(col).type)
            else:
                p, g = U64, 0
            it.append(p); fl.append(g)
        A = lambda xs: (ctypes.c_int * max(len(xs), 1))(*xs)
        self._h = lib().orc_agg_create(self.kind, len(kt), A(kt), len(ft), A(ft), A(it), A(fl))
        if not self._h:
            raise RuntimeError("orc_agg_create failed")

    def next(self, batch: pa.RecordBatch):
        if self._h is None:
            self._init(batch.schema)
        keep = []
        keys = (_Col * max(len(self.groupby_cols), 1))(*[_col(batch.column(batch.schema.get_field_index(c)), keep)
                                                        for c in self.groupby_cols])
        ins = []
        for f, col, _ in self.funcs:
            ins.append(_col(batch.column(batch.schema.get_field_index(col)), keep) if col else _Col())
        ins = (_Col * max(len(ins), 1))(*ins)
        lib().orc_agg_next(self._h, batch.num_rows, keys, ins)

    def result(self) -> pa.RecordBatch:
        L = lib()
        n = L.orc_agg_ngroups(self._h)
        names, arrays = [], []
        for c in self.agg_cols:
            j = self.groupby_cols.index(c)
            vals = np.zeros(n, np.uint64); valid = np.zeros(n, np.uint8)
            L.orc_agg_keys(self._h, j, vals.ctypes.data, valid.ctypes.data)
            t = self._schema.field(c).type
            pt, _ = physical_type(t)
            if pt == F64: v = vals.view(np.float64)
            elif pt == F32: v = vals.astype(np.uint32).view(np.float32)
            else: v = vals.astype(_NP[pt])  # truncation restores the native pattern
            names.append(c); arrays.append(_arrow_from(v, valid, t))
        for i, (f, col, out) in enumerate(self.funcs):
            cell = np.zeros((n, 2), np.uint64); valid = np.zeros(n, np.uint8)
            kind = L.orc_agg_func(self._h, i, cell.ctypes.data, valid.ctypes.data)
            in_t = self._schema.field(col).type if col else pa.uint64()
            lo = np.ascontiguousarray(cell[:, 0])
            if f in (MIN, MAX):
                arr = _minmax_out(lo, valid, in_t)
            elif kind == OUT_U64:
                arr = _arrow_from(lo, valid, pa.uint64())
            elif kind == OUT_I64:
                t = pa.int64()
                if f == SUM and (pa.types.is_time64(in_t) or pa.types.is_duration(in_t)):
                    t = in_t  # agg_func_factory.cpp:138-149 (duration: sane type)
                arr = _arrow_from(lo.view(np.int64), valid, t)
            elif kind == OUT_I32:
                arr = _arrow_from(lo.astype(np.uint32).view(np.int32), valid, in_t)
            elif kind == OUT_F64:
                arr = _arrow_from(lo.view(np.float64), valid, pa.float64())
            elif kind == OUT_F32:
                arr = _arrow_from(lo.astype(np.uint32).view(np.float32), valid, pa.float32())
            else:
                arr = _arrow_from(cell, valid, pa.decimal128(38, 0))
            names.append(out); arrays.append(arr)
        return pa.RecordBatch.from_arrays(arrays, names=names)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_agg_destroy(self._h)
            self._h = None


def cmp_mask(arr: pa.Array, op: int, literal) -> np.ndarray:
    """NumPy-semantics comparison of a column with a Python scalar -> bool ndarray."""
    keep = []
    c = _col(arr, keep)
    mask = np.zeros(len(arr), np.uint8)
    is_f = isinstance(literal, float)
    lib().orc_cmp_mask(ctypes.byref(c), op, int(is_f), float(literal), int(literal) if not is_f else 0,
                       int(arr.null_count > 0), mask.ctypes.data)
    return mask.astype(bool)


def filter_batch(batch: pa.RecordBatch, mask: np.ndarray, mask_valid=None) -> pa.RecordBatch:
    """RecordBatch.filter(mask, null_selection_behavior='emit_null') (record_batch.py:85-90)."""
    L = lib()
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    mv = None if mask_valid is None else np.ascontiguousarray(mask_valid, dtype=np.uint8)
    arrays = []
    for arr in batch.columns:
        keep = []
        c = _col(arr, keep)
        pt, _ = physical_type(arr.type)
        out = np.zeros(len(arr), _NP[pt]); ov = np.zeros(len(arr), np.uint8)
        k = L.orc_filter_col(ctypes.byref(c), m.ctypes.data, None if mv is None else mv.ctypes.data,
                             out.ctypes.data, ov.ctypes.data)
        arrays.append(_arrow_from(out[:k], ov[:k], arr.type))
    return pa.RecordBatch.from_arrays(arrays, names=batch.schema.names)


GENERIC = 3   # GenericHashAggregate: a class of the reference's pybind module, not a kind of the C restatement


def _is_numeric_type(t: pa.DataType) -> bool:        # vinum/core/aggregate.py:63-66
    return pa.types.is_integer(t) or pa.types.is_floating(t) or pa.types.is_temporal(t)


class OracleGenericAggregate:
    """TEST INFRASTRUCTURE (pure Python row loops: small inputs only).  Restates what the reference does with NON-NUMERIC data:

    * group keys of any type -- GenericHashAggregate keys its map on vectors of arrow::Scalar, equal when every scalar Equals
      (a NULL scalar equals a NULL scalar): generic_hash_aggregate.h:10-45, generic_hash_aggregate.cpp:14-40.  Restated as:
      every non-numeric key value is replaced by the index of its first appearance (a Python dict: value equality), NULL stays
      NULL, and the coded rows go through the C restatement of Single/MultiNumericalHashAggregate -- which groups by value
      equality with NULL as a group of its own, i.e. the same partition of the rows;
    * COUNT over a non-numeric column -- CountFunc only consults IsNull (agg_funcs.h:129-161): an int8 column with the same
      validity is counted instead;
    * MIN / MAX over strings -- StringMinMaxFunc, agg_funcs.h:219-261: NULL rows are skipped, a group's state starts NULL,
      `if ((row_val < last_view) ^ is_max) last = row_val` with string_view's byte-wise operator<.  Restated row by row.
    kind: ONE_GROUP / SINGLE / MULTI / GENERIC (the reference's Single and Multi classes accept string aggregate inputs too:
    TEST_F Single_/Multi_Int64Grp_StringArgFuncs, hash_agg_test.cpp:866-884)."""

    def __init__(self, kind, groupby_cols, agg_cols, funcs):
        self.kind, self.groupby, self.agg_cols, self.funcs = kind, list(groupby_cols), list(agg_cols), [tuple(f) for f in funcs]
        self._inner = None
        self._codes = {}          # key column -> {python value: code}
        self._values = {}         # key column -> [python value per code]
        self._types = {}
        self._str = {}            # (group key tuple) -> [state per string function]
        self._str_funcs = []

    def _init(self, schema):
        for c in self.groupby:
            if schema.get_field_index(c) < 0:
                raise RuntimeError("Column not found: " + c)
            if not _is_numeric_type(schema.field(c).type):
                self._codes[c], self._values[c], self._types[c] = {}, [], schema.field(c).type
        self._stand_in = set()
        self._schema_types = {f.name: f.type for f in schema}
        for i, (f, col, out) in enumerate(self.funcs):
            if col and not _is_numeric_type(schema.field(col).type):
                if f == COUNT:
                    self._stand_in.add(col)
                elif f in (MIN, MAX):
                    self._str_funcs.append(i)
                else:
                    raise RuntimeError("Column data type is not supported by " + ("sum()." if f == SUM else "avg()."))
        inner_kind = ONE_GROUP if not self.groupby else (SINGLE if len(self.groupby) == 1 and self.kind != MULTI else MULTI)
        self._numeric_funcs = [fn for i, fn in enumerate(self.funcs) if i not in self._str_funcs]
        self._inner = OracleAggregate(inner_kind, self.groupby, self.groupby, self._numeric_funcs)

    def next(self, batch: pa.RecordBatch):
        if self._inner is None:
            self._init(batch.schema)
        n = batch.num_rows
        arrays, names = [], []
        key_rows = []
        for name in batch.schema.names:
            col = batch.column(batch.schema.get_field_index(name))
            if name in self._codes:
                codes, table, vals = [], self._codes[name], self._values[name]
                for v in col.to_pylist():
                    if v is None:
                        codes.append(None)
                    else:
                        if v not in table:
                            table[v] = len(vals)
                            vals.append(v)
                        codes.append(table[v])
                col = pa.array(codes, type=pa.int32())
            elif name in self._stand_in:
                col = pa.array([0 if ok else None for ok in col.is_valid().to_pylist()], type=pa.int8())
            elif not _is_numeric_type(col.type):
                continue
            arrays.append(col); names.append(name)
        coded = pa.RecordBatch.from_arrays(arrays, names=names)
        self._inner.next(coded)
        if self._str_funcs:
            keys = [coded.column(coded.schema.get_field_index(c)) for c in self.groupby]
            # group identity as the numeric classes see it: the key's bit pattern, NULL its own group
            kb = [[(None if v is None else (float(v).hex() if isinstance(v, float) else v)) for v in k.to_pylist()] for k in keys]
            ins = {i: batch.column(batch.schema.get_field_index(self.funcs[i][1])).to_pylist() for i in self._str_funcs}
            for r in range(n):
                g = tuple(k[r] for k in kb)
                st = self._str.setdefault(g, [None] * len(self._str_funcs))       # Init: NULL unless the first row has a value
                for j, i in enumerate(self._str_funcs):
                    v = ins[i][r]
                    if v is None:
                        continue                                                   # NextIfNull: skipped
                    vb = v.encode() if isinstance(v, str) else bytes(v)
                    if st[j] is None:
                        st[j] = (vb, v)
                    elif (vb < st[j][0]) ^ (self.funcs[i][0] == MAX):
                        st[j] = (vb, v)

    def result(self) -> pa.RecordBatch:
        res = self._inner.result()
        nk = len(self.groupby)
        rows = res.num_rows
        key_cols = [res.column(j) for j in range(nk)]
        kb = [[(None if v is None else (float(v).hex() if isinstance(v, float) else v)) for v in k.to_pylist()] for k in key_cols]
        out_arrays, out_names = [], []
        for c in self.agg_cols:
            col = res.column(res.schema.get_field_index(c))
            if c in self._codes:
                vals = self._values[c]
                col = pa.array([None if v is None else vals[v] for v in col.to_pylist()], type=self._types[c])
            out_arrays.append(col); out_names.append(c)
        ni = 0
        for i, (f, colname, out) in enumerate(self.funcs):
            if i in self._str_funcs:
                j = self._str_funcs.index(i)
                vals = []
                for r in range(rows):
                    st = self._str.get(tuple(k[r] for k in kb), [None] * len(self._str_funcs))[j]
                    vals.append(None if st is None else st[1])
                out_arrays.append(pa.array(vals, type=self._in_type(colname)))
            else:
                out_arrays.append(res.column(nk + ni)); ni += 1
            out_names.append(out)
        return pa.RecordBatch.from_arrays(out_arrays, names=out_names)

    def _in_type(self, colname):
        return self._schema_types[colname]


class OracleSort:
    """Sort.next/sorted (sort.cpp:11-63): buffer batches, stable multi-key sort, take all columns."""

    def __init__(self, cols, orders):
        self.cols, self.orders, self.batches = list(cols), list(orders), []

    def next(self, batch: pa.RecordBatch):
        self.batches.append(batch)

    def sort_indices(self, table: pa.Table) -> np.ndarray:
        keep = []
        cols = (_Col * len(self.cols))(*[_col(table.column(c), keep) for c in self.cols])
        desc = (ctypes.c_int * len(self.cols))(*self.orders)
        idx = np.zeros(table.num_rows, np.int64)
        lib().orc_sort_indices(len(self.cols), cols, desc, table.num_rows, idx.ctypes.data)
        return idx

    def sorted(self) -> pa.RecordBatch:
        table = pa.Table.from_batches(self.batches).combine_chunks()
        if not all(_is_numeric_type(f.type) for f in table.schema):
            # Tables with non-numeric columns (strings, booleans, decimals ...): sort.cpp:22-44 is two calls into a third-party
            # library absent from /root/reference -- Apache Arrow (3.0.0 pinned, setup.py:33; 25 here): SortIndices over the sort
            # keys (stable, NULLs last in both directions, NaN after every number) + Take of every column.  Restated with the same
            # two calls of that library's Python binding; pinned by tests/golden/sortmix_* (outputs of the reference's own Sort).
            import pyarrow.compute as pc
            idx = pc.sort_indices(table, sort_keys=[(c, "descending" if o else "ascending") for c, o in zip(self.cols, self.orders)])
            return table.take(idx).combine_chunks().to_batches()[0] if table.num_rows else table.schema.empty_table().to_batches()[0] \
                if table.schema.empty_table().to_batches() else pa.RecordBatch.from_pylist([], schema=table.schema)
        idx = self.sort_indices(table)
        arrays = []
        for name in table.schema.names:
            keep = []
            arr = table.column(name)
            arr = arr.chunk(0) if arr.num_chunks == 1 else pa.concat_arrays(arr.chunks) if arr.num_chunks else pa.array([], arr.type)
            c = _col(arr, keep)
            pt, _ = physical_type(arr.type)
            out = np.zeros(len(idx), _NP[pt]); ov = np.zeros(len(idx), np.uint8)
            lib().orc_take_col(ctypes.byref(c), idx.ctypes.data, len(idx), out.ctypes.data, ov.ctypes.data)
            arrays.append(_arrow_from(out, ov, arr.type))
        return pa.RecordBatch.from_arrays(arrays, names=table.schema.names)
