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
import os

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


def _is_numeric(t: pa.DataType) -> bool:        # vinum/core/aggregate.py:63-66
    return pa.types.is_integer(t) or pa.types.is_floating(t) or pa.types.is_temporal(t)


def _canonical_order(batch: pa.RecordBatch, key_names):
    """Row order by the key columns' (NULL flag, bit pattern): the same for two results over the same groups whatever order their
    operators emitted them in (group identity is the key's bit pattern -- -0.0 and +0.0 are two groups, NULL is one)."""
    import numpy as np
    if batch.num_rows <= 1 or not key_names:
        return np.arange(batch.num_rows)
    keys = []
    for k in key_names:
        a = batch.column(batch.schema.names.index(k))
        valid = np.ones(len(a), bool) if a.null_count == 0 else a.is_valid().to_numpy(zero_copy_only=False)
        try:
            fixed = (not _abi_generic_key(a.type)) and a.type.bit_width in (8, 16, 32, 64)
        except (ValueError, AttributeError):
            fixed = False
        if not fixed:
            # strings / binaries / bools / decimals (group keys the library encodes itself since round 6): ranks of the distinct values
            import pyarrow.compute as pc
            enc = (a.combine_chunks() if isinstance(a, pa.ChunkedArray) else a).dictionary_encode()
            d = enc.dictionary
            if pa.types.is_string(d.type) or pa.types.is_large_string(d.type):
                d = d.cast(pa.large_binary() if pa.types.is_large_string(d.type) else pa.binary())
            order = pc.sort_indices(d).to_numpy(zero_copy_only=False) if len(d) else np.zeros(0, np.int64)
            rank_of = np.empty(len(d), np.uint64)
            rank_of[order] = np.arange(len(d), dtype=np.uint64)
            idx = enc.indices.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int64)
            raw = rank_of[idx] if len(d) else np.zeros(len(a), np.uint64)
            keys.append((~valid).astype(np.uint8))
            keys.append(np.where(valid, raw, 0))
            continue
        w = a.type.bit_width // 8
        raw = a.view({1: pa.uint8(), 2: pa.uint16(), 4: pa.uint32(), 8: pa.uint64()}[w]).fill_null(0).to_numpy(zero_copy_only=False)
        keys.append((~valid).astype(np.uint8))
        keys.append(np.where(valid, raw.astype(np.uint64), 0))
    return np.lexsort(keys[::-1])


def _abi_generic_key(t) -> bool:
    """non-numeric key types vnm_agg_op_* encode below the C ABI (vnm_arrow.cpp: GenKey)"""
    import os
    if os.environ.get("VNM_GENERIC_BELOW_ABI", "1") == "0":
        return False
    return (pa.types.is_string(t) or pa.types.is_large_string(t) or pa.types.is_binary(t) or pa.types.is_large_binary(t) or
            pa.types.is_boolean(t) or (pa.types.is_decimal(t) and t.bit_width == 128))


class _StringMinMax:
    """MIN / MAX of string / binary columns (StringMinMaxFunc, agg_funcs.h:219-261: NULLs skipped, all-NULL group -> NULL, byte-wise
    `row < last`), for every operator class (the reference runs Single_/Multi_/Generic_Int64Grp_StringArgFuncs,
    hash_agg_test.cpp:866-894).  The GPU aggregates order-preserving RANKS: per batch the column is dictionary-encoded (on the
    device, vnm_strdict_encode) and the batch's dictionary is ranked there (vnm_strdict_ranks_device), and a numeric operator of the caller's class takes MIN / MAX of the int32 ranks per
    group; the winners go back to strings -- one candidate per group, batch and function.  result() ranks the candidates of all
    batches against each other and takes MIN / MAX once more.  String comparisons happen once per distinct value (the
    dictionary's sort on the device), not once per row."""

    def __init__(self, op_cls, groupby_cols, funcs):
        self._cls, self._groupby = op_cls, list(groupby_cols)
        self._funcs = list(funcs)                       # AggFuncDefs (MIN / MAX over non-numeric columns)
        self._partials = []

    def _aggregate(self, key_arrays, rank_arrays, names_of):
        """one numeric operator over (keys, int32 rank columns): MIN / MAX per function"""
        defs = [AggFuncDef(f.func, names_of[i], f"r{i}") for i, f in enumerate(self._funcs)]
        op = self._cls(defs) if self._cls is OneGroupAggregate else self._cls(self._groupby, self._groupby, defs)
        cols = list(key_arrays) + list(rank_arrays.values())
        names = list(self._groupby) + list(rank_arrays.keys())
        op.next(pa.RecordBatch.from_arrays(cols, names=names))
        return op.result()

    @staticmethod
    def _ranks(column):
        """(int32 ranks with the column's NULLs, sorted dictionary): rank order == byte-wise string order"""
        import numpy as np
        import pyarrow.compute as pc
        if isinstance(column, pa.ChunkedArray):
            column = column.combine_chunks()
        t = column.type
        if len(column) and (pa.types.is_string(t) or pa.types.is_large_string(t) or pa.types.is_binary(t) or pa.types.is_large_binary(t)):
            # the batch's dictionary from the device (vnm_strdict_encode, a fresh handle: every value of the batch is new to it);
            # the host sorts the DISTINCT values and turns codes into ranks (one gather)
            # ... and ranks them there too (round 5, vnm_strdict_ranks_device: the distinct values sorted byte-wise by the multi-key radix
            # sort over their 8-byte chunks); the host only gathers rank[code] and lines the dictionary up by rank
            from .device import DeviceBuffer
            kd = KeyDictionary(t)
            codes = kd.encode(column)
            d = pa.concat_arrays(kd._chunks) if kd._chunks else pa.array([], type=t)
            c = codes.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int64)
            mask = ~codes.is_valid().to_numpy(zero_copy_only=False) if codes.null_count else None
            if not len(d):
                return pa.array(np.zeros(len(c), np.int32), type=pa.int32(), mask=mask), d
            top = int(L.lib().vnm_strdict_ids(kd._h))
            dev = DeviceBuffer(max(top, 1) * 4)
            L.check(L.lib().vnm_strdict_ranks_device(kd._h, dev.ptr, None))
            rank_of_id = dev.to_host(np.int32, top)               # (ids never handed out: garbage, never looked at)
            live = np.nonzero(kd._pos[:top] >= 0)[0]
            by_rank = np.empty(len(d), np.int64)
            by_rank[rank_of_id[live]] = kd._pos[live]              # rank -> position in `d`
            return pa.array(rank_of_id[c], type=pa.int32(), mask=mask), d.take(pa.array(by_rank))
        enc = column.dictionary_encode()
        d = enc.dictionary
        if pa.types.is_string(d.type) or pa.types.is_large_string(d.type):
            d_sort = d.cast(pa.large_binary() if pa.types.is_large_string(d.type) else pa.binary())   # byte-wise, as string_view's operator<
        else:
            d_sort = d
        order = pc.sort_indices(d_sort).to_numpy(zero_copy_only=False)
        rank_of = np.empty(len(d), np.int32)
        rank_of[order] = np.arange(len(d), dtype=np.int32)
        idx = enc.indices
        ranks = rank_of[idx.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int64)] if len(d) else np.zeros(len(idx), np.int32)
        mask = ~idx.is_valid().to_numpy(zero_copy_only=False) if idx.null_count else None
        return pa.array(ranks, type=pa.int32(), mask=mask), d.take(pa.array(order))

    def next(self, key_arrays, batch: pa.RecordBatch) -> None:
        # (the rank columns get names of their own: the aggregated column may be a group-by column as well, and a batch with two
        #  columns of one name has no field-by-name)
        ranks, dicts, names_of, rank_name = {}, {}, [], {}
        for f in self._funcs:
            c = f.column_name
            if c not in rank_name:
                rank_name[c] = f"__vnm_rank{len(rank_name)}"
                ranks[rank_name[c]], dicts[c] = self._ranks(batch.column(batch.schema.names.index(c)))
            names_of.append(rank_name[c])
        res = self._aggregate(key_arrays, ranks, names_of)
        nk = len(self._groupby)
        cand = [dicts[f.column_name].take(res.column(nk + i)) for i, f in enumerate(self._funcs)]     # NULL rank -> NULL string
        self._partials.append(pa.RecordBatch.from_arrays([res.column(j) for j in range(nk)] + cand,
                                                         names=self._groupby + [f"s{i}" for i in range(len(self._funcs))]))
        # a long stream (the reference's default batch is 10 000 rows) must not keep one partial result per batch until result()
        # (VERDICT r05 weak #11): every _FOLD_EVERY batches the candidates are merged into ONE partial -- G rows x functions stay
        if len(self._partials) >= self._FOLD_EVERY:
            self._partials = [self.result()]

    _FOLD_EVERY = 8

    def result(self) -> pa.RecordBatch:
        """(keys..., one string column per function), one row per group"""
        t = pa.Table.from_batches(self._partials).combine_chunks()
        nk = len(self._groupby)
        if len(self._partials) == 1:
            return t.to_batches()[0]
        # candidates of all batches against each other: one shared sorted dictionary per function, ranks, MIN / MAX again
        ranks, dicts, names_of = {}, {}, []
        for i in range(len(self._funcs)):
            ranks[f"c{i}"], dicts[f"c{i}"] = self._ranks(t.column(nk + i))
            names_of.append(f"c{i}")
        res = self._aggregate([t.column(j).combine_chunks() for j in range(nk)], ranks, names_of)
        cand = [dicts[f"c{i}"].take(res.column(nk + i)) for i in range(len(self._funcs))]
        return pa.RecordBatch.from_arrays([res.column(j) for j in range(nk)] + cand,
                                          names=self._groupby + [f"s{i}" for i in range(len(self._funcs))])


class _HashAggregateBase:
    _KIND = None

    def __init__(self, groupby_cols, agg_cols, agg_funcs):
        self._groupby, self._agg_cols, self._funcs = list(groupby_cols), list(agg_cols), list(agg_funcs)
        self._h = None
        self._strmm = None          # _StringMinMax for MIN / MAX over non-numeric columns
        self._stand_in = set()      # non-numeric columns that are only COUNTed: an int8 column with the same validity stands in
        self._abi_str = set()       # string / binary columns under MIN / MAX: passed to the library as they are (round 6)
        self._create(self._groupby, self._agg_cols, self._funcs)

    def _create(self, groupby_cols, agg_cols, agg_funcs):
        lib = L.lib()
        ftypes = (ctypes.c_int * max(len(agg_funcs), 1))(*[int(f.func) for f in agg_funcs])
        self._h = lib.vnm_agg_op_create(self._KIND, len(groupby_cols), _cstrs(groupby_cols), len(agg_cols),
                                        _cstrs(agg_cols), len(agg_funcs), ftypes,
                                        _cstrs([f.column_name for f in agg_funcs]),
                                        _cstrs([f.out_col_name for f in agg_funcs]))
        if not self._h:
            raise RuntimeError(L.last_error())

    def _first_batch(self, batch: pa.RecordBatch) -> None:
        """The first batch fixes the schema (base_aggregate.cpp:91-119).  Functions over NON-NUMERIC columns: COUNT counts through
        an int8 stand-in with the column's validity; MIN / MAX of strings / binaries go to _StringMinMax; SUM / AVG raise what the
        reference raises (agg_func_factory.cpp:174,245)."""
        schema = batch.schema
        str_funcs = []
        for f in self._funcs:
            c = f.column_name
            if not c or c not in schema.names or _is_numeric(schema.field(c).type):
                continue
            t = schema.field(c).type
            if f.func == AggFuncType.COUNT:
                self._stand_in.add(c)
            elif f.func in (AggFuncType.MIN, AggFuncType.MAX) and (pa.types.is_string(t) or pa.types.is_large_string(t) or pa.types.is_binary(t)
                                                                  or pa.types.is_large_binary(t)):
                # round 6: vnm_agg_op_* take such columns themselves (one growing device dictionary per column, candidates per group
                # in HBM); the shim's own route stays reachable for A / B runs
                if os.environ.get("VNM_STRING_MINMAX_IN_SHIM"):
                    str_funcs.append(f)
                else:
                    self._abi_str.add(c)
            else:
                raise RuntimeError({AggFuncType.MIN: "Column data type is not supported by min()/max().",
                                    AggFuncType.MAX: "Column data type is not supported by min()/max().",
                                    AggFuncType.SUM: "Column data type is not supported by sum().",
                                    AggFuncType.AVG: "Column data type is not supported by avg()."}[f.func])
        if str_funcs:
            for c in self._groupby:
                if c not in schema.names:
                    raise RuntimeError(f"Column not found: {c}")
            # the numeric operator keeps every other function and emits ALL group-by columns: the two results are lined up by key
            L.lib().vnm_agg_op_destroy(self._h)
            self._h = None
            rest = [f for f in self._funcs if f not in str_funcs]
            self._create(self._groupby, self._groupby, rest)
            self._strmm = _StringMinMax(type(self), self._groupby, str_funcs)

    def _numeric_view(self, batch: pa.RecordBatch) -> pa.RecordBatch:
        """what crosses the boundary: numeric columns as they are, COUNT-only non-numeric columns as their int8 stand-ins; group-by
        columns of the types the library dictionary-encodes itself (round 6: vnm_agg_op_* take utf8 / binary / bool / decimal128 keys)
        as they are"""
        import numpy as np
        if not self._stand_in and self._strmm is None and not self._abi_str:
            return batch
        arrays, names = [], []
        for i, name in enumerate(batch.schema.names):
            col = batch.column(i)
            if name in self._groupby and _abi_generic_key(col.type):
                pass
            elif name in self._abi_str:
                pass                                      # MIN / MAX of a string / binary column: the library's own (round 6; a COUNT over it as well)
            elif name in self._stand_in:
                col = pa.array(np.zeros(len(col), np.int8), mask=(~col.is_valid().to_numpy(zero_copy_only=False)) if col.null_count else None)
            elif not _is_numeric(col.type):
                continue
            arrays.append(col)
            names.append(name)
        return pa.RecordBatch.from_arrays(arrays, names=names)

    # The reference streams 10 000-row batches by default (vinum/__init__.py:52).  Handing each of them to the library costs ~50 us
    # of Python + Arrow C export per batch -- 20x what its rows cost on the device -- so batches below 2^20 rows are kept here
    # (a list append) and cross the boundary together, as one Arrow C stream, once 2^22 rows are waiting, or at result(); the
    # library keeps the small batches of a stream as they are and stages them to the device as one.
    _SMALL_ROWS = 1 << 20
    _FLUSH_ROWS = 1 << 22

    def _send(self, batch: pa.RecordBatch) -> None:
        if self._strmm is not None:
            self._strmm.next([batch.column(batch.schema.names.index(c)) for c in self._groupby], batch)
        batch = self._numeric_view(batch)
        c = _CStructs()
        batch._export_to_c(c.arr_ptr, c.sch_ptr)
        if L.lib().vnm_agg_op_next(self._h, c.arr_ptr, c.sch_ptr) != 0:
            raise RuntimeError(L.last_error())

    def _flush(self) -> None:
        pending, self._pending, self._pending_rows = getattr(self, "_pending", None), [], 0
        if not pending:
            return
        if self._stand_in or self._strmm is not None or len(pending) == 1:
            # (stand-in / rank columns are built per batch: one joined batch, as before)
            joined = pa.Table.from_batches(pending).combine_chunks()
            for b in joined.to_batches():
                self._send(b)
            return
        # the waiting batches cross the boundary as ONE Arrow C stream: no concatenation on the host (that copy was most of
        # the cost: 0.8 GB per 5e7 rows), one call instead of one per batch; the library stages them to the device together
        reader = pa.RecordBatchReader.from_batches(pending[0].schema, pending)
        stream = ctypes.create_string_buffer(40)      # struct ArrowArrayStream: five pointers
        reader._export_to_c(ctypes.addressof(stream))
        if L.lib().vnm_agg_op_next_stream(self._h, ctypes.addressof(stream)) != 0:
            raise RuntimeError(L.last_error())

    def next(self, batch: pa.RecordBatch) -> None:
        if not hasattr(self, "_pending"):
            # the FIRST batch goes straight through: it fixes the schema, and an unknown column or an unsupported type must raise
            # from this call as it does in the reference (base_aggregate.cpp:91-131)
            self._pending, self._pending_rows = [], 0
            self._schema0 = batch.schema
            self._first_batch(batch)
            self._send(batch)
            return
        # a batch that can raise (any schema other than the first batch's) or that is large crosses the boundary in THIS call:
        # what the reference raises from the offending Next() must not surface from a later next() or from result()
        if batch.num_rows >= self._SMALL_ROWS or batch.schema != self._schema0:
            self._flush()
            self._send(batch)
            return
        self._pending.append(batch)
        self._pending_rows += batch.num_rows
        if self._pending_rows >= self._FLUSH_ROWS:
            self._flush()

    def result(self) -> pa.RecordBatch:
        self._flush()
        c = _CStructs()
        if L.lib().vnm_agg_op_result(self._h, c.arr_ptr, c.sch_ptr) != 0:
            raise RuntimeError(L.last_error())
        res = pa.RecordBatch._import_from_c(c.arr_ptr, c.sch_ptr)
        return res if self._strmm is None else self._with_strings(res)

    def _with_strings(self, res: pa.RecordBatch) -> pa.RecordBatch:
        """BaseAggregate::Result's column order (base_aggregate.cpp:52-60): the selected group-by columns, then EVERY function in
        AggFuncDef order -- the numeric ones from the device operator, the string MIN / MAX from _StringMinMax, rows lined up by
        the group keys."""
        if not hasattr(self, "_pending"):            # no batch at all: nothing to line up (the numeric result is what there is)
            return res
        smm = self._strmm.result()
        o_num, o_str = _canonical_order(res, self._groupby), _canonical_order(smm, self._groupby)
        if len(o_num) != len(o_str):
            raise RuntimeError(f"aggregate: the numeric result holds {len(o_num)} groups, the string MIN / MAX result {len(o_str)} (internal error)")
        res, smm = res.take(pa.array(o_num)), smm.take(pa.array(o_str))
        # ... and the SAME groups: after the canonical order the key columns of both results must agree in validity and bits, or the
        # string columns would be attached to the wrong rows without anybody noticing
        for k in self._groupby:
            if k not in res.schema.names:
                continue
            a, b = res.column(res.schema.names.index(k)), smm.column(smm.schema.names.index(k))
            same = a.is_valid().equals(b.is_valid())
            if same and len(a):
                try:
                    bw = a.type.bit_width
                except (ValueError, AttributeError):      # (a string / binary key column: compared by value)
                    bw = 0
                ut = {8: pa.uint8(), 16: pa.uint16(), 32: pa.uint32(), 64: pa.uint64()}.get(bw if not pa.types.is_boolean(a.type) else 0)
                if a.type != b.type:
                    same = False
                elif ut is None:
                    same = a.equals(b)
                else:
                    same = a.view(ut).fill_null(0).equals(b.view(ut).fill_null(0))      # (bit patterns: NaN keys compare by payload)
            if not same:
                raise RuntimeError(f"aggregate: the numeric and the string MIN / MAX results disagree on the groups of key '{k}' (internal error)")
        arrays, names = [], []
        for c in self._agg_cols:
            arrays.append(res.column(res.schema.names.index(c))); names.append(c)
        nk = len(self._groupby)
        for f in self._funcs:
            if f in self._strmm._funcs:
                arrays.append(smm.column(nk + self._strmm._funcs.index(f)))
            else:
                arrays.append(res.column(res.schema.names.index(f.out_col_name)))
            names.append(f.out_col_name)
        return pa.RecordBatch.from_arrays(arrays, names=names)

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
    """Running dictionary of one non-numeric group-by column: value -> int32 code.

    The reference keys its generic map on arrow::Scalar vectors (generic_hash_aggregate.h:10-45): hash + Equals per row.
    Here every batch is encoded once on ingest and the GPU groups by 4-byte codes with the numeric machinery (NULL stays
    NULL: its own group, as in the reference where a NULL scalar equals a NULL scalar).

    utf8 / large_utf8 / binary / large_binary columns are encoded ON THE DEVICE (`vnm_strdict_encode`, csrc/vnm_strdict.hip:
    offsets and bytes cross PCIe once, one kernel hashes every row and finds or inserts it in the dictionary table; only the
    values a batch ADDS come back, and this object keeps them to decode the result's key column).  The host route it replaces
    (Arrow's `dictionary_encode` + a NumPy merge of the batch's dictionary into the running one) ran at 7-17 M rows/s, 1.5 M
    rows/s at 5e6 distinct values.  Other types (bool, decimal, date64 ...: a handful of values or fixed width) keep it."""

    def __init__(self, arrow_type: pa.DataType):
        self.type = arrow_type
        self.values = pa.array([], type=arrow_type)     # host route: the dictionary in order of first appearance
        self._device = (pa.types.is_string(arrow_type) or pa.types.is_large_string(arrow_type) or pa.types.is_binary(arrow_type)
                        or pa.types.is_large_binary(arrow_type))
        self._h = None
        self._chunks = []          # device route: the values each batch added ...
        self._pos = None           # ... and id -> position in their concatenation (-1: an id never handed out)
        self._n_values = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.lib().vnm_strdict_destroy(h)
            except Exception:
                pass

    def handle(self):
        """the device dictionary (vnm_strdict*) behind a utf8 / binary column, created on first use"""
        if self._h is None:
            self._h = L.lib().vnm_strdict_create()
            if not self._h:
                raise RuntimeError(L.last_error())
        return self._h

    def absorb_new(self):
        """after an encode through this dictionary's handle (vnm_strdict_encode here, vnm_csv_parse_block_ex in io.py): fetch the values
        it added -- the host copy of the dictionary is the concatenation of these"""
        import numpy as np
        lib = L.lib()
        n_new, new_bytes = ctypes.c_int64(0), ctypes.c_int64(0)
        L.check(lib.vnm_strdict_last_new(self._h, ctypes.byref(n_new), ctypes.byref(new_bytes)))
        if not n_new.value:
            return
        wide = pa.types.is_large_string(self.type) or pa.types.is_large_binary(self.type)
        ids = np.empty(n_new.value, np.int32)
        lens = np.empty(n_new.value, np.int32)
        data = np.empty(max(new_bytes.value, 1), np.uint8)
        L.check(lib.vnm_strdict_fetch_new(self._h, ids.ctypes.data, lens.ctypes.data, data.ctypes.data))
        offs = np.zeros(n_new.value + 1, np.int64 if wide else np.int32)
        np.cumsum(lens, out=offs[1:])
        self._chunks.append(pa.Array.from_buffers(self.type, n_new.value, [None, pa.py_buffer(offs), pa.py_buffer(data[:new_bytes.value])]))
        top = int(lib.vnm_strdict_ids(self._h))
        if self._pos is None or len(self._pos) < top:
            grown = np.full(max(top, 2 * (len(self._pos) if self._pos is not None else 0)), -1, np.int64)
            if self._pos is not None:
                grown[:len(self._pos)] = self._pos
            self._pos = grown
        self._pos[ids] = self._n_values + np.arange(n_new.value, dtype=np.int64)
        self._n_values += n_new.value

    def _encode_device(self, column: pa.Array) -> pa.Array:
        import numpy as np
        lib = L.lib()
        n = len(column)
        wide = pa.types.is_large_string(self.type) or pa.types.is_large_binary(self.type)
        vbuf, obuf, dbuf = column.buffers()
        codes = np.empty(max(n, 1), np.int32)
        L.check(lib.vnm_strdict_encode(self.handle(), obuf.address if obuf is not None else None, 1 if wide else 0,
                                       dbuf.address if dbuf is not None and dbuf.size else None,
                                       vbuf.address if (vbuf is not None and column.null_count) else None,
                                       column.offset, n, codes.ctypes.data, None, None, None))
        self.absorb_new()
        codes = codes[:n]
        mask = (codes < 0) if column.null_count else None
        return pa.array(codes, type=pa.int32(), mask=mask)

    def encode(self, column) -> pa.Array:
        import numpy as np
        import pyarrow.compute as pc
        if isinstance(column, pa.ChunkedArray):
            column = column.combine_chunks()
        if self._device and len(column):
            return self._encode_device(column)
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

    def code_of(self, value):
        """The code of `value` in this dictionary, or None (predicates on a dictionary-coded column: `city = 'Berlin'` is a
        comparison of int32 codes in HBM)."""
        import pyarrow.compute as pc
        vals = self.values_by_code()
        if not len(vals):
            return None
        i = pc.index(vals, pa.scalar(value, type=self.type)).as_py()
        return None if i < 0 else int(i)

    def rank_bounds(self, value):
        """(number of dictionary values < value, 1 if value is in the dictionary else 0): with r = rank_column's rank of a row,
        v < value <=> r < less, v <= value <=> r < less + present, v > value <=> r >= less + present, v >= value <=> r >= less
        (byte-wise order, as Arrow compares utf8 / binary)."""
        import pyarrow.compute as pc
        vals = self.values_by_code()
        if not len(vals):
            return 0, 0
        lit = pa.scalar(value, type=self.type)
        less = int(pc.sum(pc.less(vals, lit)).as_py() or 0)
        return less, 0 if self.code_of(value) is None else 1

    def rank_column(self, codes_col):
        """ORDER BY a dictionary-coded column: the rows' order-preserving RANKS as an int32 DeviceColumn (same validity as the
        codes: NULL stays NULL).  utf8 / binary dictionaries are ranked on the device (vnm_strdict_ranks_device: the distinct values
        sorted byte-wise, as Arrow's SortIndices compares them); the small host-route dictionaries (decimals, date64 ...) by Arrow."""
        import numpy as np
        from .device import DeviceBuffer, DeviceColumn
        lib = L.lib()
        n = codes_col.length
        if self._device and self._h is not None:
            top = max(int(lib.vnm_strdict_ids(self._h)), 1)
            rank_of = DeviceBuffer(top * 4)
            L.check(lib.vnm_strdict_ranks_device(self._h, rank_of.ptr, None))
        else:
            import pyarrow.compute as pc
            order = pc.sort_indices(self.values).to_numpy(zero_copy_only=False) if len(self.values) else np.zeros(0, np.int64)
            r = np.zeros(max(len(self.values), 1), np.int32)
            r[order] = np.arange(len(order), dtype=np.int32)
            rank_of = DeviceBuffer.from_host(r)
        off = codes_col.offset if codes_col._validity is not None else 0      # (one Arrow offset serves values and bitmap)
        out = DeviceBuffer(max(off + n, 1) * 4)
        L.check(lib.vnm_strdict_codes_to_ranks(codes_col.values_ptr + codes_col.offset * 4, rank_of.ptr, n, out.ptr + off * 4, None))
        return DeviceColumn(out, codes_col._validity, off, n, pa.int32(), keep=(rank_of, codes_col))

    def values_by_code(self) -> pa.Array:
        """The dictionary as an array indexed by CODE (a code the device never handed out: NULL) -- what the ranks exchange to build
        one dictionary before partial groups keyed by these codes can travel (distributed.union_dictionary)."""
        import numpy as np
        if self._device and self._h is not None:
            # (cached until the dictionary grows: a predicate on a dictionary column asks once per batch and literal -- ADVICE r05 --, and
            #  the concat + take is O(distinct values))
            top = int(L.lib().vnm_strdict_ids(self._h))
            stamp = (top, len(self._chunks))
            cached = getattr(self, "_by_code_cache", None)
            if cached is not None and cached[0] == stamp:
                return cached[1]
            values = pa.concat_arrays(self._chunks) if self._chunks else pa.array([], type=self.type)
            pos = self._pos[:top] if self._pos is not None else np.zeros(0, np.int64)
            out = values.take(pa.array(np.where(pos < 0, 0, pos), type=pa.int64(), mask=(pos < 0) if (pos < 0).any() else None))
            self._by_code_cache = (stamp, out)
            return out
        return self.values

    def decode(self, codes: pa.Array) -> pa.Array:
        if self._device and self._h is not None:
            import numpy as np
            values = pa.concat_arrays(self._chunks) if self._chunks else pa.array([], type=self.type)
            c = codes.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int64)
            pos = self._pos[c] if self._pos is not None and len(c) else np.zeros(len(c), np.int64)
            null = ~codes.is_valid().to_numpy(zero_copy_only=False) if codes.null_count else np.zeros(len(c), bool)
            if ((pos < 0) & ~null).any():
                raise RuntimeError("KeyDictionary.decode: a code that was never handed out (internal error)")
            return values.take(pa.array(np.where(null, 0, pos), type=pa.int64(), mask=null if null.any() else None))
        return self.values.take(codes)


class GenericHashAggregate:
    """GROUP BY over keys of ANY Arrow type -- strings, bools, decimals, mixed with numeric keys
    (vinum/core/vinum_lib.cpp:92-109, vinum_cpp/src/operators/aggregate/generic_hash_aggregate.{h,cpp}).

    Non-numeric key columns are dictionary-encoded on ingest (KeyDictionary) and the query runs through the numeric GPU
    operators (single key: the whole single-key machinery incl. the partitioned paths; several keys: packed composite
    keys); result() maps the codes back, so key columns keep their type.  Aggregate INPUT columns must be numeric --
    COUNT(col) of a non-numeric column counts through an int8 stand-in with the same validity and MIN / MAX of strings
    (agg_funcs.h:219-261) are ranks-on-the-device (_StringMinMax), both in the numeric classes underneath."""

    def __init__(self, groupby_cols, agg_cols, agg_funcs):
        self._groupby, self._agg_cols, self._funcs = list(groupby_cols), list(agg_cols), list(agg_funcs)
        self._inner = None
        self._dicts = {}

    def _init(self, batch: pa.RecordBatch):
        schema = batch.schema
        for c in self._groupby:
            if c not in schema.names:
                raise RuntimeError(f"Column not found: {c}")       # base_aggregate.cpp:121-131
            t = schema.field(c).type
            if not _is_numeric(t) and not _abi_generic_key(t):     # (strings, binaries, bools, decimal128: the library's own dictionary since round 6)
                self._dicts[c] = KeyDictionary(t)
        # (functions over non-numeric columns -- COUNT, string MIN / MAX -- are the numeric classes' business: _first_batch)
        # A non-numeric column that is group key AND aggregate input (`SELECT city, min(city), count(city) ... GROUP BY city`,
        # generic_hash_aggregate.h:10-45 + StringMinMaxFunc agg_funcs.h:219-261 take it): the key travels as dictionary codes under its
        # own name, the functions read the column itself under a second name
        self._alias = {f.column_name: "__vnm_in_" + f.column_name for f in self._funcs if f.column_name in self._dicts}
        funcs = [AggFuncDef(f.func, self._alias.get(f.column_name, f.column_name), f.out_col_name) for f in self._funcs]
        cls = SingleNumericalHashAggregate if len(self._groupby) == 1 else MultiNumericalHashAggregate
        self._inner = cls(self._groupby, self._agg_cols, funcs)

    def next(self, batch: pa.RecordBatch) -> None:
        # the first batch fixes the schema (and raises what the reference raises); later small batches are encoded together:
        # dictionary-encoding 10 000 rows at a time costs more in Python than in work
        if self._inner is None:
            self._init(batch)
            self._pending, self._pending_rows = [], 0
            self._schema0 = batch.schema
            self._encode_and_send(batch)
            return
        if batch.num_rows >= (1 << 20) or batch.schema != self._schema0:   # (a schema change raises from this call, as in the reference)
            self._flush()
            self._encode_and_send(batch)
            return
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
        fn_inputs = {f.column_name for f in self._funcs if f.column_name}
        arrays, names = [], []
        for i, name in enumerate(batch.schema.names):
            col = batch.column(i)
            if name in self._dicts:
                if name in self._alias:
                    arrays.append(col)
                    names.append(self._alias[name])
                col = self._dicts[name].encode(col)
            elif not _is_numeric(col.type) and name not in fn_inputs and name not in self._groupby:
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
    """Sort (vinum_cpp/src/operators/sort/sort.cpp:11-63): next() retains the batches; sorted() = arrow SortIndices over the
    sort keys + Take of EVERY column of the table, any Arrow type.  All of it happens below the C ABI (vnm_sort_op_*), on the
    device: numeric / temporal keys and columns; string / binary KEYS as order-preserving ranks from the device string dictionary
    (NULL stays NULL and goes last in both directions, equal values share a rank, so the stable sort leaves ties in row order
    like SortIndices); string / binary / boolean / decimal128 PAYLOAD gathered on the device; decimal128 keys as two key words.
    Boolean sort keys raise like the reference's SortOperator (vinum/core/algebra.py:191-201: Arrow 3.0 cannot sort them).
    A column of a type the library does not take (lists, structs, ...) that is NOT a sort key stays with this wrapper: an
    int64 row-id column rides along and the column is gathered with Arrow `take` by the sorted row ids."""

    @staticmethod
    def _below_the_abi(t) -> bool:
        from .device import is_supported
        return (is_supported(t) or pa.types.is_string(t) or pa.types.is_large_string(t) or pa.types.is_binary(t) or
                pa.types.is_large_binary(t) or pa.types.is_boolean(t) or (pa.types.is_decimal(t) and t.bit_width == 128))

    def __init__(self, sort_cols, sort_order):
        self._cols, self._orders = list(sort_cols), [int(o) for o in sort_order]
        self._h = self._create(self._cols, self._orders)
        self._pending, self._pending_rows = [], 0
        self._carried = None           # columns of types the library does not take: kept here, gathered by row id after the sort
        self._carried_batches = []
        self._rows_fed = 0

    @staticmethod
    def _create(cols, orders):
        ords = (ctypes.c_int * max(len(orders), 1))(*orders)
        h = L.lib().vnm_sort_op_create(len(cols), _cstrs(cols), ords)
        if not h:
            raise RuntimeError(L.last_error())
        return h

    def next(self, batch: pa.RecordBatch) -> None:
        # Sort::Next only retains the batch (sort.cpp:11-13); nothing can fail before sorted().  Small batches (the reference's
        # default is 10 000 rows) are kept here and cross the boundary joined, as in the aggregates.
        if self._carried is None:
            self._schema_order = batch.schema.names
            self._carried = [f.name for f in batch.schema if not self._below_the_abi(f.type)]
            for name in self._carried:
                if name in self._cols:
                    raise RuntimeError(f"Failed to sort table. (ORDER BY a column of type {batch.schema.field(name).type})")
        if self._carried:
            import numpy as np
            self._carried_batches.append(batch.select(self._carried))
            keep = [f.name for f in batch.schema if f.name not in self._carried]
            ids = pa.array(np.arange(self._rows_fed, self._rows_fed + batch.num_rows, dtype=np.int64))
            batch = pa.RecordBatch.from_arrays([batch.column(n) for n in keep] + [ids], names=keep + ["__vnm_row_id"])
        self._rows_fed += batch.num_rows
        if self._pending and batch.schema != self._pending[0].schema:
            self._flush()
        self._pending.append(batch)
        self._pending_rows += batch.num_rows
        if self._pending_rows >= (1 << 24):
            self._flush()

    @staticmethod
    def _feed(h, batches) -> None:
        # one Arrow C stream for all waiting batches (no concatenation on the host; Sort::Next only retains them anyway)
        reader = pa.RecordBatchReader.from_batches(batches[0].schema, batches)
        stream = ctypes.create_string_buffer(40)      # struct ArrowArrayStream: five pointers
        reader._export_to_c(ctypes.addressof(stream))
        if L.lib().vnm_sort_op_next_stream(h, ctypes.addressof(stream)) != 0:
            raise RuntimeError(L.last_error())

    def _flush(self) -> None:
        pending, self._pending, self._pending_rows = self._pending, [], 0
        if not pending:
            return
        self._feed(self._h, pending)

    @staticmethod
    def _sorted_of(h, limit) -> pa.RecordBatch:
        c = _CStructs()
        if L.lib().vnm_sort_op_sorted(h, int(limit), c.arr_ptr, c.sch_ptr) != 0:
            raise RuntimeError(L.last_error())
        return pa.RecordBatch._import_from_c(c.arr_ptr, c.sch_ptr)

    def sorted(self, limit: int = 0) -> pa.RecordBatch:
        """limit (extension, default 0 = everything): only the first `limit` rows are needed (LIMIT pushed
        into the sort; identical rows to sorting everything and slicing)."""
        self._flush()
        res = self._sorted_of(self._h, limit)
        if not self._carried:
            return res
        ids = res.column(res.schema.names.index("__vnm_row_id"))
        carried = pa.Table.from_batches(self._carried_batches).combine_chunks()
        self._carried_batches = []
        names = [n for n in res.schema.names if n != "__vnm_row_id"]
        cols = {n: res.column(res.schema.names.index(n)) for n in names}
        for n in self._carried:
            cols[n] = carried.column(n).combine_chunks().take(ids)
        order = self._schema_order if getattr(self, "_schema_order", None) else names + self._carried
        return pa.RecordBatch.from_arrays([cols[n] for n in order], names=order)

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
