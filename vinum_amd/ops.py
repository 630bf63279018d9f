"""Device-level operator wrappers over the C ABI (HBM-resident columns in, HBM-resident columns out).

These are the building blocks of the operator mirror in vinum_amd/core/ and of bench.py; they never
touch the host except for scalar results (row counts) and final result materialisation.
"""
import ctypes
import os

import numpy as np
import pyarrow as pa

from . import _lib as L
from .device import DeviceBuffer, DeviceColumn, _NP, _WIDTH, arrow_from_numpy, dcol_array, physical_type

CMP_OPS = {"==": L.EQ, "=": L.EQ, "!=": L.NE, "<>": L.NE, ">": L.GT, ">=": L.GE, "<": L.LT, "<=": L.LE}


def _stream_ptr(stream):
    if stream is None:
        return None
    return ctypes.c_void_p(int(stream))


def filter_cmp(pred: DeviceColumn, op, literal, payload, stream=None):
    """`WHERE pred <op> literal` over HBM columns: returns (compacted payload columns, rows kept).

    Replaces FilterOperator._kernel + RecordBatch.filter + the NumPy comparison lambda
    (reference: vinum/core/algebra.py:119-123, vinum/arrow/record_batch.py:85-90,
    vinum/core/expressions.py:30-36)."""
    lib = L.lib()
    op = CMP_OPS.get(op, op)
    outs = []
    n = pred.length
    for start in range(0, max(len(payload), 1), 8):  # the kernel compacts up to 8 columns per pass
        chunk = payload[start:start + 8]
        vals = [DeviceBuffer(n * _WIDTH[c.vnm_type]) for c in chunk]
        vbytes = [DeviceBuffer(n) if c.validity_ptr else None for c in chunk]
        ov = (ctypes.c_void_p * max(len(chunk), 1))(*[b.ptr for b in vals])
        ob = (ctypes.c_void_p * max(len(chunk), 1))(*[(b.ptr if b else None) for b in vbytes])
        count = ctypes.c_int64(0)
        is_f = isinstance(literal, float)
        L.check(lib.vnm_filter_cmp(ctypes.byref(pred.dcol()), op, int(is_f), float(literal), 0 if is_f else int(literal),
                                   len(chunk), dcol_array(chunk), ov, ob, ctypes.byref(count), _stream_ptr(stream)))
        k = count.value
        for c, v, b in zip(chunk, vals, vbytes):
            bitmap = None
            if b is not None:
                bitmap = DeviceBuffer((k + 7) // 8)
                L.check(lib.vnm_pack_validity(b.ptr, k, bitmap.ptr, _stream_ptr(stream)))
                L.check(lib.vnm_device_synchronize())
            outs.append(c.like(v, bitmap, 0, k))
        if not payload:
            return [], k
    return outs, (outs[0].length if outs else 0)


def filter_mask(mask: DeviceBuffer, mask_valid, length, payload, stream=None):
    """RecordBatch.filter(mask, null_selection_behavior='emit_null') with a device byte mask."""
    lib = L.lib()
    outs = []
    for start in range(0, len(payload), 8):
        chunk = payload[start:start + 8]
        vals = [DeviceBuffer(length * _WIDTH[c.vnm_type]) for c in chunk]
        need_v = [bool(c.validity_ptr) or mask_valid is not None for c in chunk]
        vbytes = [DeviceBuffer(length) if nv else None for nv in need_v]
        ov = (ctypes.c_void_p * len(chunk))(*[b.ptr for b in vals])
        ob = (ctypes.c_void_p * len(chunk))(*[(b.ptr if b else None) for b in vbytes])
        count = ctypes.c_int64(0)
        L.check(lib.vnm_filter_mask(mask.ptr, mask_valid.ptr if mask_valid is not None else None, length, len(chunk),
                                    dcol_array(chunk), ov, ob, ctypes.byref(count), _stream_ptr(stream)))
        k = count.value
        for c, v, b in zip(chunk, vals, vbytes):
            bitmap = None
            if b is not None:
                bitmap = DeviceBuffer((k + 7) // 8)
                L.check(lib.vnm_pack_validity(b.ptr, k, bitmap.ptr, _stream_ptr(stream)))
                L.check(lib.vnm_device_synchronize())
            outs.append(c.like(v, bitmap, 0, k))
    return outs, (outs[0].length if outs else 0)


def _ints(xs):
    return (ctypes.c_int * max(len(xs), 1))(*xs)


class DeviceAggregate:
    """Streaming hash aggregate over HBM-resident columns (device level of the three reference classes).

    funcs: list of (func_id, input_column_index or None, arrow_type of the input or None)."""

    def __init__(self, kind, key_types, funcs, expected_groups=0, rank_aligned=False, stream_mode=False):
        """stream_mode: record batches of a stream are only RECORDED by next() and go to the device together (vnm_agg_set_async:
        one launch of the path's kernels over all waiting batches instead of launches, allocations and a host read-back per
        batch).  This object keeps the batches' DeviceColumns alive until they have been processed."""
        self.kind = kind
        self.key_arrow = list(key_types)
        self.funcs = list(funcs)
        kt = [physical_type(t)[0] for t in key_types]
        ft, it, fl, ids = [], [], [], []
        for f, col, t in funcs:
            ft.append(f)
            if t is None:
                it.append(L.U64); fl.append(0); ids.append(-1)
            else:
                p, g = physical_type(t)
                it.append(p); fl.append(g); ids.append(col if col is not None else -1)
        self._spec = (kind, kt, ft, it, fl, ids)
        self._h = L.lib().vnm_agg_create(kind, len(kt), _ints(kt), len(ft), _ints(ft), _ints(it), _ints(fl), _ints(ids))
        if not self._h:
            raise RuntimeError(L.last_error())
        if expected_groups:
            L.check(L.lib().vnm_agg_set_hint(self._h, int(expected_groups)))
        if rank_aligned:     # the result will be exchanged between ranks (vinum_amd.distributed)
            L.check(L.lib().vnm_agg_set_exchange_mode(self._h, 1))
        self._pred = False
        self._waiting = []        # stream_mode: the columns of the batches the library has only recorded so far
        self._stream_mode = bool(stream_mode)
        if stream_mode:
            L.check(L.lib().vnm_agg_set_async(self._h, 1))

    def sync(self, stream=None):
        """stream_mode: process every waiting batch (vnm_agg_sync); their columns may go afterwards."""
        L.check(L.lib().vnm_agg_sync(self._h, _stream_ptr(stream)))
        self._waiting.clear()

    def set_predicate(self, op, literal):
        op = CMP_OPS.get(op, op)
        is_f = isinstance(literal, float)
        L.check(L.lib().vnm_agg_set_predicate(self._h, 1, op, int(is_f), float(literal), 0 if is_f else int(literal)))
        self._pred = True

    def set_input_expr(self, func_idx, expr, col_names):
        """The input of function `func_idx` (and of the functions sharing its column id) is `expr` -- a prefix expression
        over `col_names` -- instead of a column: `sum((1 - total) * (2 + tax))`.  Evaluated in registers inside the scan in
        the hot shape, otherwise materialised by one fused projection pass (vnm_agg_set_input_expr)."""
        prog = compile_expr(expr, {n: i for i, n in enumerate(col_names)})
        L.check(L.lib().vnm_agg_set_input_expr(self._h, int(func_idx), len(prog), prog, len(col_names)))

    def next(self, keys, inputs, pred=None, nrows=None, stream=None, expr_cols=None):
        """keys: list[DeviceColumn]; inputs: one DeviceColumn (or None for COUNT(*) / expression inputs) per function;
        expr_cols: the columns of the expression set with set_input_expr, in its col_names order."""
        if nrows is None:
            nrows = keys[0].length if keys else next(c.length for c in inputs if c is not None) if any(
                c is not None for c in inputs) else 0
        p = ctypes.byref(pred.dcol()) if pred is not None else None
        if expr_cols is not None:
            L.check(L.lib().vnm_agg_next_device_expr(self._h, nrows, dcol_array(keys), dcol_array(inputs), p, len(expr_cols),
                                                     dcol_array(expr_cols), _stream_ptr(stream)))
            return
        L.check(L.lib().vnm_agg_next_device(self._h, nrows, dcol_array(keys), dcol_array(inputs), p, _stream_ptr(stream)))
        if self._stream_mode:
            # The library reads the buffers of a RECORDED batch when the waiting batches are processed: they are kept until then, and
            # no longer -- a stream larger than HBM must not stay resident until result() (the reference streams such inputs batch by
            # batch, vinum/api/stream_reader.py:32-94).  vnm_agg_waiting names the oldest call whose batch still waits.
            oldest, last = ctypes.c_int64(-1), ctypes.c_int64(-1)
            L.check(L.lib().vnm_agg_waiting(self._h, None, None, ctypes.byref(oldest), ctypes.byref(last)))
            if oldest.value < 0:
                self._waiting.clear()
            else:
                self._waiting.append((last.value, keys, inputs, pred))
                if self._waiting[0][0] < oldest.value:
                    self._waiting = [w for w in self._waiting if w[0] >= oldest.value]

    def waiting(self):
        """(batches, rows) the library holds recorded (stream mode)."""
        nb, nr = ctypes.c_int64(0), ctypes.c_int64(0)
        L.check(L.lib().vnm_agg_waiting(self._h, ctypes.byref(nb), ctypes.byref(nr), None, None))
        return nb.value, nr.value

    def finish(self, stream=None) -> int:
        n = ctypes.c_int64(0)
        L.check(L.lib().vnm_agg_finish(self._h, ctypes.byref(n), _stream_ptr(stream)))
        self._waiting.clear()
        return n.value

    def layout(self):
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        L.check(L.lib().vnm_agg_layout(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def word_layout(self):
        """How this operator's functions lower onto accumulator words (vnm_agg_plan_host): {'n_key_words', 'merge':
        merge kind per word, 'ops': [(update kind, distinct input column, word)]}."""
        kind, kt, ft, it, fl, ids = self._spec
        nkw, nw, nops = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        mk = (ctypes.c_int * 40)()
        op3 = (ctypes.c_int * (3 * 48))()
        L.check(L.load().vnm_agg_plan_host(kind, len(kt), _ints(kt), len(ft), _ints(ft), _ints(it), _ints(fl), _ints(ids),
                                           ctypes.byref(nkw), ctypes.byref(nw), mk, ctypes.byref(nops), op3))
        return {"n_key_words": nkw.value, "merge": [mk[w] for w in range(nw.value)],
                "ops": [(op3[3 * o], op3[3 * o + 1], op3[3 * o + 2]) for o in range(nops.value)]}

    def dense_ptrs(self):
        kw, aw = self.layout()
        K = (ctypes.c_void_p * max(kw, 1))()
        A = (ctypes.c_void_p * max(aw, 1))()
        L.check(L.lib().vnm_agg_dense_ptrs(self._h, K, A))
        return [K[i] for i in range(kw)], [A[i] for i in range(aw)]

    def bucket_by_owner(self, world, out_rows_ptr, stream=None):
        """Rows of the finished run grouped by owner rank (multi-GPU exchange); returns the per-owner counts."""
        counts = (ctypes.c_int64 * world)()
        L.check(L.lib().vnm_agg_bucket_by_owner(self._h, world, out_rows_ptr, counts, _stream_ptr(stream)))
        return list(counts)

    def merge(self, n, key_ptrs, acc_ptrs, stream=None):
        K = (ctypes.c_void_p * max(len(key_ptrs), 1))(*key_ptrs)
        A = (ctypes.c_void_p * max(len(acc_ptrs), 1))(*acc_ptrs)
        L.check(L.lib().vnm_agg_merge_device(self._h, n, K, A, _stream_ptr(stream)))

    def run_partitions(self) -> int:
        """F > 0 when the finished result is a partition-structured run (partition-aligned exchange possible)."""
        return int(L.lib().vnm_agg_run_partitions(self._h))

    def run_reorder(self, world, rows_ptr, part_counts_ptr, stream=None):
        counts = (ctypes.c_int64 * world)()
        L.check(L.lib().vnm_agg_run_reorder(self._h, world, rows_ptr, part_counts_ptr, counts, _stream_ptr(stream)))
        return list(counts)

    def merge_partitioned(self, world, nlocal, rows_ptr, src_row_offsets, part_counts_ptr, stream=None):
        """False when a merged partition does not fit the LDS merge table (ranks holding disjoint key sets): the
        handle is still empty and the caller falls back to merge_rows on the same rows."""
        offs = (ctypes.c_int64 * (world + 1))(*src_row_offsets)
        rc = L.lib().vnm_agg_merge_partitioned(self._h, world, nlocal, rows_ptr, offs, part_counts_ptr, _stream_ptr(stream))
        if rc == 2:
            return False
        L.check(rc)
        return True

    def merge_rows(self, n, rows_ptr, stream=None):
        L.check(L.lib().vnm_agg_merge_rows(self._h, n, rows_ptr, _stream_ptr(stream)))

    def merge_row_blocks(self, nblocks, block_rows, blocks_ptr, stream=None):
        """blocks of (block_rows + 1) rows with a header row each (distributed.exchange_small_fixed)"""
        L.check(L.lib().vnm_agg_merge_row_blocks(self._h, int(nblocks), int(block_rows), blocks_ptr, _stream_ptr(stream)))

    def result_arrays(self, key_indices, out_names_keys, out_names_funcs) -> pa.RecordBatch:
        """Column order follows BaseAggregate::Result (base_aggregate.cpp:47-68): selected group keys
        first, then the functions; output types follow agg_func_factory.cpp:13-329."""
        lib = L.lib()
        n = self.finish()
        names, arrays = [], []
        for j, name in zip(key_indices, out_names_keys):
            vals = np.zeros(max(n, 1), np.uint64)
            valid = np.zeros(max(n, 1), np.uint8)
            L.check(lib.vnm_agg_result_key(self._h, j, vals.ctypes.data, valid.ctypes.data))
            vals, valid = vals[:n], valid[:n]
            t = self.key_arrow[j]
            pt, _ = physical_type(t)
            if pt == L.F64: v = vals.view(np.float64)
            elif pt == L.F32: v = vals.astype(np.uint32).view(np.float32)
            else: v = vals.astype(_NP[pt])
            names.append(name)
            arrays.append(arrow_from_numpy(v, ~valid.astype(bool), t))
        for i, ((f, col, in_t), name) in enumerate(zip(self.funcs, out_names_funcs)):
            cells = np.zeros((max(n, 1), 2), np.uint64)
            valid = np.zeros(max(n, 1), np.uint8)
            kind = ctypes.c_int(0)
            L.check(lib.vnm_agg_result_func(self._h, i, cells.ctypes.data, valid.ctypes.data, ctypes.byref(kind)))
            cells, valid = cells[:n], valid[:n]
            names.append(name)
            arrays.append(_func_array(f, in_t, kind.value, cells, valid))
        return pa.RecordBatch.from_arrays(arrays, names=names)

    def estimate_groups(self, key, nrows, stream=None) -> int:
        """The operator's own group-count estimate for a batch, without aggregating it (vnm_agg_estimate_groups); 0 = not
        applicable.  distributed.agree_on_group_count() makes every rank use the same value."""
        est = ctypes.c_int64(0)
        L.check(L.lib().vnm_agg_estimate_groups(self._h, int(nrows), ctypes.byref(key.dcol()), ctypes.byref(est), _stream_ptr(stream)))
        return est.value

    def set_hint(self, expected_groups):
        L.check(L.lib().vnm_agg_set_hint(self._h, int(expected_groups)))

    def result_device(self, key_indices=None, stream=None):
        """BaseAggregate::Result with the columns LEFT IN HBM: selected group keys first, then the functions
        (base_aggregate.cpp:47-68), finalised by a device kernel straight into Arrow-layout buffers (typed values +
        validity bitmap) -- no accumulator words cross PCIe.  Returns DeviceColumns.  An int64 / uint64 SUM that
        overflows 64 bits (-> decimal128, agg_funcs.h:366-389) raises NeedsHostFinalize: use result_arrays()."""
        lib = L.lib()
        if key_indices is None:
            key_indices = range(len(self.key_arrow))
        which = [~j for j in key_indices] + list(range(len(self.funcs)))
        nc = len(which)
        if nc == 0:
            return []
        # the library allocates the columns: the group count is only known once the last pass has run, and a pending
        # dense-path final pass writes the result columns itself (vnm_agg_result_device_alloc)
        c_which = (ctypes.c_int * nc)(*which)
        c_vals = (ctypes.c_void_p * nc)()
        c_bitmaps = (ctypes.c_void_p * nc)()
        kinds = (ctypes.c_int * nc)()
        nulls = (ctypes.c_int64 * nc)()
        ng = ctypes.c_int64(0)
        rc = lib.vnm_agg_result_device_alloc(self._h, nc, c_which, c_vals, c_bitmaps, kinds, nulls, ctypes.byref(ng), _stream_ptr(stream))
        if rc == 2:
            raise NeedsHostFinalize(L.last_error() or "a 64-bit SUM overflowed: decimal128 result")
        L.check(rc)
        self._waiting.clear()
        n = ng.value
        self.result_rows = n
        cols = []
        for c, w in enumerate(which):
            t = self.key_arrow[~w] if w < 0 else _func_arrow_type(self.funcs[w][0], self.funcs[w][2], kinds[c])
            vals = DeviceBuffer.adopt(c_vals[c], max(n, 1) * 8)
            bitmap = DeviceBuffer.adopt(c_bitmaps[c], ((n + 63) // 64) * 8) if c_bitmaps[c] else None
            cols.append(DeviceColumn(vals, bitmap if nulls[c] else None, 0, n, t))
        return cols

    # -- multi-GPU: the dense-key path with a code range all ranks agree on (vinum_amd.distributed.exchange_dense_tables) --
    def dense_range(self, key, nrows, stream=None):
        """(lo, hi) of this rank's sampled key range as order-preserving unsigned images; lo > hi: no dense path."""
        lo, hi = ctypes.c_uint64(0), ctypes.c_uint64(0)
        L.check(L.lib().vnm_agg_dense_range(self._h, int(nrows), ctypes.byref(key.dcol()), ctypes.byref(lo), ctypes.byref(hi),
                                            _stream_ptr(stream)))
        return lo.value, hi.value

    def set_dense_range(self, lo, hi):
        L.check(L.lib().vnm_agg_set_dense_range(self._h, physical_type(self.key_arrow[0])[0], int(lo), int(hi)))

    def dense_table(self, stream=None):
        """(ptr, bits, geometry) of the direct-addressed tables of the pending final pass, or None."""
        ptr, bits = ctypes.c_void_p(0), ctypes.c_int(0)
        geo = (ctypes.c_uint64 * 4)()
        L.check(L.lib().vnm_agg_dense_table(self._h, ctypes.byref(ptr), ctypes.byref(bits), geo, _stream_ptr(stream)))
        self._waiting.clear()
        if not ptr.value:
            return None
        return ptr.value, bits.value, tuple(int(g) for g in geo)

    def merge_dense_tables(self, like, slice_ptrs, code0, n, stream=None):
        P = (ctypes.c_void_p * len(slice_ptrs))(*slice_ptrs)
        L.check(L.lib().vnm_agg_merge_dense_tables(self._h, like._h, len(slice_ptrs), P, int(code0), int(n), _stream_ptr(stream)))

    def close(self):
        if getattr(self, "_h", None):
            L.lib().vnm_agg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NeedsHostFinalize(RuntimeError):
    """The result column changes type on the host (int64 SUM -> decimal128): use DeviceAggregate.result_arrays()."""


def _func_arrow_type(f, in_t, kind) -> pa.DataType:
    """Output type of one aggregate function (agg_func_factory.cpp: COUNT -> uint64 :31-34; MIN/MAX type-preserving
    :35-107; SUM :108-176; AVG :177-247)."""
    if f in (L.MIN, L.MAX):
        return in_t
    if kind == L.OUT_U64:
        return pa.uint64()
    if kind == L.OUT_I64:
        if f == L.SUM and in_t is not None and (pa.types.is_time64(in_t) or pa.types.is_duration(in_t)):
            return in_t
        return pa.int64()
    if kind == L.OUT_I32:
        return in_t
    if kind == L.OUT_F64:
        return pa.float64()
    if kind == L.OUT_F32:
        return pa.float32()
    return pa.decimal128(38, 0)


def _func_array(f, in_t, kind, cells, valid) -> pa.Array:
    """Result column of one aggregate function with the reference's output type
    (agg_func_factory.cpp: COUNT -> uint64 :31-34; MIN/MAX type-preserving :35-107; SUM :108-176;
    AVG :177-247)."""
    mask = ~valid.astype(bool)
    lo = np.ascontiguousarray(cells[:, 0])
    if f in (L.MIN, L.MAX):
        pt, _ = physical_type(in_t)
        if pt in (L.F32, L.F64): v = lo.view(np.float64).astype(_NP[pt])
        elif pt in (L.U8, L.U16, L.U32, L.U64): v = lo.astype(_NP[pt])
        else: v = lo.view(np.int64).astype(_NP[pt])
        return arrow_from_numpy(v, mask, in_t)
    if kind == L.OUT_U64:
        return arrow_from_numpy(lo, mask, pa.uint64())
    if kind == L.OUT_I64:
        t = pa.int64()
        if f == L.SUM and in_t is not None and (pa.types.is_time64(in_t) or pa.types.is_duration(in_t)):
            t = in_t
        return arrow_from_numpy(lo.view(np.int64), mask, t)
    if kind == L.OUT_I32:
        return arrow_from_numpy(lo.astype(np.uint32).view(np.int32), mask, in_t)
    if kind == L.OUT_F64:
        return arrow_from_numpy(lo.view(np.float64), mask, pa.float64())
    if kind == L.OUT_F32:
        return arrow_from_numpy(lo.astype(np.uint32).view(np.float32), mask, pa.float32())
    # decimal128(38, 0): 16-byte little-endian two's complement cells are Arrow's native layout
    buf = pa.py_buffer(np.ascontiguousarray(cells).tobytes())
    vbuf = None
    if mask.any():
        vbuf = pa.py_buffer(np.packbits(valid.astype(bool), bitorder="little").tobytes())
    return pa.Array.from_buffers(pa.decimal128(38, 0), len(valid), [vbuf, buf], null_count=int(mask.sum()))


# ---- sort / take --------------------------------------------------------------------------------------
def sort_indices(keys, orders, limit=0, stream=None) -> DeviceBuffer:
    """arrow::compute::SortIndices as used by Sort::Sorted (vinum_cpp/src/operators/sort/sort.cpp:22-37):
    stable multi-key, NaN after values and NULL after NaN for both directions.  Returns int64 row ids in HBM
    (the first `limit` entries when limit > 0)."""
    n = keys[0].length
    out = DeviceBuffer(max(n, 1) * 8)
    od = (ctypes.c_int * len(orders))(*orders)
    L.check(L.lib().vnm_sort_indices(len(keys), dcol_array(keys), od, n, int(limit), out.ptr, _stream_ptr(stream)))
    return out


def sort_indices_keyed(keys, orders, stream=None):
    """Full sort that also hands back the SORTED FIRST KEY when the sort can rebuild it from its codes (int64 / uint64 /
    float64 without NULL, NaN or -0.0): (row ids, DeviceColumn or None).  The caller then gathers (`take`) only the other
    columns -- a gather is 25 ms per 1e9 rows and column, half of what the sort itself costs."""
    n = keys[0].length
    out = DeviceBuffer(max(n, 1) * 8)
    keybuf = DeviceBuffer(max(n, 1) * 8)
    wrote = ctypes.c_int(0)
    od = (ctypes.c_int * len(orders))(*orders)
    L.check(L.lib().vnm_sort_indices_keyed(len(keys), dcol_array(keys), od, n, 0, out.ptr, keybuf.ptr, ctypes.byref(wrote),
                                           _stream_ptr(stream)))
    if not wrote.value:
        return out, None
    return out, keys[0].like(keybuf, None, 0, n)


def take(col: DeviceColumn, indices: DeviceBuffer, n, stream=None) -> DeviceColumn:
    """compute::Take for one column (sort.cpp:40)."""
    lib = L.lib()
    vals = DeviceBuffer(max(n, 1) * _WIDTH[col.vnm_type])
    vb = DeviceBuffer(max(n, 1)) if col.validity_ptr else None
    L.check(lib.vnm_take(ctypes.byref(col.dcol()), indices.ptr, n, vals.ptr, vb.ptr if vb else None, _stream_ptr(stream)))
    bitmap = None
    if vb is not None:
        bitmap = DeviceBuffer((n + 7) // 8)
        L.check(lib.vnm_pack_validity(vb.ptr, n, bitmap.ptr, _stream_ptr(stream)))
        L.check(lib.vnm_device_synchronize())
    return col.like(vals, bitmap, 0, n)


# ---- projection ------------------------------------------------------------------------------------------
_EX = {"add": L.EX_ADD, "sub": L.EX_SUB, "mul": L.EX_MUL, "div": L.EX_DIV, "mod": L.EX_MOD, "neg": L.EX_NEG,
       "band": L.EX_BAND, "bor": L.EX_BOR, "bxor": L.EX_BXOR, "bnot": L.EX_BNOT,
       "+": L.EX_ADD, "-": L.EX_SUB, "*": L.EX_MUL, "/": L.EX_DIV, "%": L.EX_MOD,
       "eq": L.EX_EQ, "ne": L.EX_NE, "gt": L.EX_GT, "ge": L.EX_GE, "lt": L.EX_LT, "le": L.EX_LE,
       "==": L.EX_EQ, "=": L.EX_EQ, "!=": L.EX_NE, "<>": L.EX_NE, ">": L.EX_GT, ">=": L.EX_GE, "<": L.EX_LT, "<=": L.EX_LE,
       "and": L.EX_AND, "or": L.EX_OR, "not": L.EX_NOT}


_ARROW_OF = {L.I8: pa.int8(), L.I16: pa.int16(), L.I32: pa.int32(), L.I64: pa.int64(), L.U8: pa.uint8(), L.U16: pa.uint16(),
             L.U32: pa.uint32(), L.U64: pa.uint64(), L.F32: pa.float32(), L.F64: pa.float64()}


def _desugar(e):
    """BETWEEN / IN rewrite onto the primitive predicates, exactly what the reference's callables compute:
    between -> logical_and(x >= low, x <= high), not_between -> logical_or(x < low, x > high)
    (vinum/core/expressions.py:43-48); in / not_in -> np.isin(x, values[, invert]) (:39-40)."""
    if not isinstance(e, tuple):
        return e
    if e[0] == "strong":
        return e
    op, args = e[0], [_desugar(x) for x in e[1:]]
    if op == "between":
        return ("and", ("ge", args[0], args[1]), ("le", args[0], args[2]))
    if op == "not_between":
        return ("or", ("lt", args[0], args[1]), ("gt", args[0], args[2]))
    if op in ("in", "not_in"):
        vals = list(e[2])
        if not vals:
            raise ValueError("IN with an empty list")
        # np.isin turns the list into an ARRAY (int64 / float64: strong types), unlike a bare literal operand
        if any(isinstance(v, float) for v in vals):
            vals = [float(v) for v in vals]
        vals = [("strong", v) for v in vals]
        if op == "in":
            return ("or",) + tuple(("eq", args[0], v) for v in vals) if len(vals) > 1 else ("eq", args[0], vals[0])
        return ("and",) + tuple(("ne", args[0], v) for v in vals) if len(vals) > 1 else ("ne", args[0], vals[0])
    return (op,) + tuple(args)


def _emit_expr(expr, col_index, out):
    """Append the postfix form of one nested prefix expression to `out` (tuples op, arg, imm_f, imm_i)."""

    def emit(e):
        if isinstance(e, str):
            out.append((L.EX_COL, col_index[e], 0.0, 0))
        elif isinstance(e, bool):
            raise TypeError("boolean literals are not arithmetic operands")
        elif isinstance(e, int):
            out.append((L.EX_CONST_I, 0, 0.0, int(e)))
        elif isinstance(e, float):
            out.append((L.EX_CONST_F, 0, float(e), 0))
        elif e[0] == "strong":
            out.append((L.EX_CONST_F, 1, float(e[1]), 0) if isinstance(e[1], float) else (L.EX_CONST_I, 1, 0.0, int(e[1])))
        elif e[0] in ("is_null", "is_not_null"):
            out.append((L.EX_IS_NULL if e[0] == "is_null" else L.EX_IS_NOT_NULL, col_index[e[1]], 0.0, 0))
        else:
            op = _EX[e[0]]
            args = e[1:]
            if op in (L.EX_NEG, L.EX_BNOT, L.EX_NOT):
                emit(args[0])
                out.append((op, 0, 0.0, 0))
            else:
                emit(args[0])
                for a in args[1:]:
                    emit(a)
                    out.append((op, 0, 0.0, 0))
    emit(_desugar(expr))


def _program(ins):
    prog = (L.ExprIns * len(ins))()
    for i, (op, arg, f, k) in enumerate(ins):
        prog[i].op, prog[i].arg, prog[i].imm_f, prog[i].imm_i = op, arg, f, k
    return prog


def compile_expr(expr, col_index):
    """Nested prefix expression -> postfix vnm_expr_ins list.
    expr := column name | int | float | (op, expr[, expr...]); n-ary chains fold left like
    VectorizedExpression._apply_binary_args_function (vinum/core/base.py:145-151)."""
    out = []
    _emit_expr(expr, col_index, out)
    return _program(out)


def project(expr, columns: dict, length=None, stream=None) -> DeviceColumn:
    """One fused kernel per output expression (replaces the per-node NumPy ufunc passes of
    vinum/core/expressions.py:13-24).  columns: name -> DeviceColumn."""
    names = list(columns)
    cols = [columns[n] for n in names]
    if length is None:
        length = cols[0].length if cols else 1
    prog = compile_expr(expr, {n: i for i, n in enumerate(names)})
    out = DeviceBuffer(max(length, 1) * 8)
    ot = ctypes.c_int(0)
    L.check(L.lib().vnm_project(len(prog), prog, len(cols), dcol_array(cols), length, out.ptr, ctypes.byref(ot),
                                _stream_ptr(stream)))
    if ot.value == L.MASK_U8:
        return DeviceColumn(out, None, 0, length, pa.uint8())   # byte mask (predicate program)
    return DeviceColumn(out, None, 0, length, _ARROW_OF[ot.value])


def project_many(exprs, columns: dict, length=None, stream=None):
    """A whole SELECT list in ONE kernel (vnm_project_multi): every input column is read from HBM once, every
    output written once -- ProjectOperator._kernel's loop over expressions (vinum/core/algebra.py:52-64) without
    its per-node temporaries.  Returns one DeviceColumn per expression."""
    exprs = list(exprs)
    if not exprs:
        return []
    if length is None:
        length = next(iter(columns.values())).length if columns else 1
    # pack expressions into programs that respect the kernel limits (64 instructions, 16 columns, 16 outputs)
    chunks, cur, cur_cols, cur_ins = [], [], [], 0
    for k, e in enumerate(exprs):
        tmp = []
        ecols = columns_of(e)
        _emit_expr(e, {n: 0 for n in ecols}, tmp)
        n_ins = len(tmp) + 1
        merged = cur_cols + [c for c in ecols if c not in cur_cols]
        if cur and (cur_ins + n_ins > 64 or len(merged) > 16 or len(cur) >= 16):
            chunks.append((cur, cur_cols))
            cur, cur_ins, merged = [], 0, list(ecols)
        cur.append(k)
        cur_cols, cur_ins = merged, cur_ins + n_ins
    chunks.append((cur, cur_cols))
    result = [None] * len(exprs)
    for ks, names in chunks:
        cols = [columns[n] for n in names]
        index = {n: i for i, n in enumerate(names)}
        ins = []
        for j, k in enumerate(ks):
            _emit_expr(exprs[k], index, ins)
            ins.append((L.EX_STORE, j, 0.0, 0))
        prog = _program(ins)
        bufs = [DeviceBuffer(max(length, 1) * 8) for _ in ks]
        ptrs = (ctypes.c_void_p * len(ks))(*[b.ptr for b in bufs])
        types = (ctypes.c_int * len(ks))()
        L.check(L.lib().vnm_project_multi(len(prog), prog, len(cols), dcol_array(cols), length, len(ks), ptrs, types,
                                          _stream_ptr(stream)))
        for k, b, t in zip(ks, bufs, types):
            at = pa.uint8() if t == L.MASK_U8 else _ARROW_OF[t]
            result[k] = DeviceColumn(b, None, 0, length, at)
    return result


def columns_of(expr):
    """Column names referenced by a (possibly sugared) expression, in first-use order."""
    seen = []

    def walk(e):
        if isinstance(e, str):
            if e not in seen:
                seen.append(e)
        elif isinstance(e, tuple):
            if e[0] == "lit":            # ("lit", "Berlin"): a string / bytes LITERAL (a bare str is a column name)
                return
            if e[0] in ("is_null", "is_not_null"):
                walk(e[1])
            elif e[0] in ("in", "not_in"):
                walk(e[1])
            else:
                for x in e[1:]:
                    walk(x)
    walk(expr)
    return seen


def predicate_mask(expr, columns: dict, length, stream=None) -> DeviceBuffer:
    """Evaluate a boolean expression tree (comparisons, AND/OR/NOT, IS [NOT] NULL, BETWEEN, IN) into a
    device byte mask -- the masks the reference builds with NumPy / pyarrow.compute one node at a time
    (vinum/core/expressions.py:27-48).  NULL operands compare False (NaN), `!=` True."""
    col = project(expr, {n: columns[n] for n in columns_of(expr)}, length=length, stream=stream)
    if col.arrow_type != pa.uint8():
        raise TypeError("WHERE expects a boolean expression")
    return col._values
