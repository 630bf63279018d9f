"""CSV ingest on the GPU: the device counterpart of stream_csv() / read_csv() (vinum/io/arrow.py:9-61, 64-110), which are
thin wrappers around pyarrow.csv.open_csv / read_csv and parse on one CPU core.

`GpuCsvReader` reads the file in blocks that end on a row boundary, hands each block's TEXT to the library
(`vnm_csv_parse_block`: one H2D copy, newline scan, one lane per row, exact decimal -> float64) and yields
`DeviceRecordBatch`es whose numeric columns were never Arrow arrays in host memory.  The schema (names, which columns are
int64 / float64) comes from pyarrow's own type inference over the first block, as `pyarrow.csv.open_csv` does it.  Only the
columns the query needs are parsed.  What the device parser does not handle is handed to pyarrow for that block: columns of
other types (dictionary-encoded as usual afterwards), fields outside the exact parser's domain ("nan", > 19
significant digits, a quoted number with an escaped quote ...) -- per column and per block, never silently guessed.  Quoted fields are
handled on the device since round 4 (delimiters inside quotes do not split a row); only a row that ENDS inside a quote (a newline
in a value) sends its block to pyarrow.

Round 5: `string`, `date32`, `timestamp[s]` and `timestamp[ns]` columns are parsed on the device as well
(`vnm_csv_parse_block_ex`): a string column is born as the int32 codes of its running dictionary (the field bytes are hashed where
they lie in the staged text), dates and timestamps by integer calendar arithmetic.  What is left to pyarrow per column and block:
zone-aware timestamps, other ISO 8601 spellings, strings with an escaped quote (booleans -- as codes of [false, true] -- and
times of day hh:mm[:ss] are parsed on the device as well).
"""
import ctypes
import io
from typing import List, Optional, Sequence

import pyarrow as pa
import pyarrow.csv as pacsv

from . import _lib as L
from .core.base import DeviceRecordBatch
from .device import DeviceBuffer, DeviceColumn


class _OwnedColumn(DeviceColumn):
    """A column whose HBM buffers belong to the library's pool (vnm_csv_parse_block): freed with vnm_free_column."""

    def __init__(self, dcol: L.DCol, arrow_type):
        super().__init__(dcol.values, dcol.validity or None, int(dcol.offset), int(dcol.length), arrow_type)
        self._dcol = dcol

    def __del__(self):
        try:
            if self._dcol is not None:
                L.lib().vnm_free_column(ctypes.byref(self._dcol))
                self._dcol = None
        except Exception:
            pass


class GpuCsvReader:
    """Streaming CSV reader with the protocol FileReaderOperator expects (`read_next_batch`, StopIteration at the end),
    plus `read_next_device_batch()` for the GPU operators.  columns: the column names the query touches (None = all)."""

    def __init__(self, source, columns: Optional[Sequence[str]] = None, block_size: int = 64 << 20, numeric_only: bool = False):
        self._numeric_only = bool(numeric_only)     # True: the round-4 split (strings, dates, timestamps through pyarrow) -- for comparisons
        self._f = open(source, "rb", buffering=0) if isinstance(source, (str, bytes)) else source
        self._block = int(block_size)
        # one reusable buffer: [carry from the previous block | freshly read bytes]; blocks are handed to the library as
        # (pointer, length) views of it -- the text is never copied on the host
        self._buf = bytearray(self._block + (1 << 20))
        self._have = 0            # valid bytes in the buffer
        self._eof = False
        self._fill()
        if self._have == 0:
            raise pa.ArrowInvalid("CSV parse error: Empty CSV file")
        nl = self._buf.find(b"\n", 0, self._have)
        if nl < 0:
            raise pa.ArrowInvalid("CSV parse error: no line end in the first block")
        self._header = bytes(self._buf[: nl + 1])
        self._start = nl + 1      # first unconsumed byte of the buffer
        # pyarrow's own header parsing and type inference over the first rows fix names and types for the stream (open_csv
        # infers from its first block too)
        end = self._buf.rfind(b"\n", self._start, min(self._have, self._start + (1 << 20))) + 1
        probe = pacsv.read_csv(io.BytesIO(self._header + bytes(self._buf[self._start:max(end, self._start)])),
                               read_options=pacsv.ReadOptions(use_threads=False))
        self.schema = probe.schema
        self._names = list(probe.schema.names)
        self._want = list(columns) if columns is not None else list(self._names)
        for c in self._want:
            if c not in self._names:
                raise ValueError(f'Column "{c}" is not found.')
        self._dicts = {}
        if not self._numeric_only:
            for n in self._want:          # (boolean columns: the dictionary [false, true] must exist before any block, whichever parser reads it)
                if self.schema.field(n).type == pa.bool_():
                    self._dictionary(n)

    # -- block reading: every block handed on ends with a newline ----------------------------------------------------
    def _fill(self):
        """read until the buffer is full or the file ends"""
        mv = memoryview(self._buf)
        while not self._eof and self._have < len(self._buf) - (1 << 16):
            got = self._f.readinto(mv[self._have:])
            if not got:
                self._eof = True
                if self._have and self._buf[self._have - 1] != 0x0A:      # a last line without a newline
                    self._buf[self._have] = 0x0A
                    self._have += 1
                break
            self._have += got

    def _next_text(self):
        """(offset, length) of the next block inside the buffer: whole rows only"""
        if self._start >= self._have:
            if self._eof:
                raise StopIteration
            self._have, self._start = 0, 0
            self._fill()
            if self._have == 0:
                raise StopIteration
        cut = self._buf.rfind(b"\n", self._start, self._have)
        if cut < 0:
            if self._eof:
                raise StopIteration
            raise pa.ArrowInvalid("CSV parse error: a row longer than the block size")
        off, length = self._start, cut + 1 - self._start
        self._pending_tail = (cut + 1, self._have)
        return off, length

    def _advance(self):
        """after a block was consumed: move the partial last row to the front and read on"""
        lo, hi = self._pending_tail
        tail = hi - lo
        self._buf[:tail] = self._buf[lo:hi]
        self._have, self._start = tail, 0
        if not self._eof:
            self._fill()
        elif tail == 0:
            self._start = self._have = 0

    # -- parsing --------------------------------------------------------------------------------------------------------
    def _host_parse(self, text: bytes, names: List[str]) -> pa.Table:
        opts = pacsv.ConvertOptions(include_columns=names, column_types={n: self.schema.field(n).type for n in names})
        return pacsv.read_csv(io.BytesIO(self._header + text), read_options=pacsv.ReadOptions(use_threads=False), convert_options=opts)

    def read_next_device_batch(self) -> DeviceRecordBatch:
        off, length = self._next_text()
        try:
            return self._parse_block(off, length)
        finally:
            self._advance()

    def _device_type(self, name: str):
        """the library's parser for this column's inferred Arrow type, or None (pyarrow parses the column)"""
        t = self.schema.field(name).type
        if t == pa.int64():
            return L.I64
        if t == pa.float64():
            return L.F64
        if self._numeric_only:
            return None
        if t == pa.string():
            return L.CSV_STRING
        if t == pa.date32():
            return L.CSV_DATE32
        if t == pa.timestamp("s"):
            return L.CSV_TIMESTAMP_S
        if t == pa.timestamp("ns"):
            return L.CSV_TIMESTAMP_NS
        if t == pa.bool_():
            return L.CSV_BOOL
        if t == pa.time32("s"):
            return L.CSV_TIME32_S
        return None

    def _dictionary(self, name: str):
        from .vinum_lib import KeyDictionary
        if name not in self._dicts:
            t = self.schema.field(name).type
            self._dicts[name] = KeyDictionary(t)
            if t == pa.bool_():      # the device parser writes 0 / 1: the codes of [false, true], whatever value a file shows first
                self._dicts[name].values = pa.array([False, True])
        return self._dicts[name]

    def _parse_block(self, off: int, length: int) -> DeviceRecordBatch:
        lib = L.lib()
        text_ptr = ctypes.addressof((ctypes.c_char * length).from_buffer(self._buf, off))
        gpu_cols = [n for n in self._want if self._device_type(n) is not None]
        gpu_cols.sort(key=self._names.index)
        cols = {}
        nrows = None
        host_cols = [n for n in self._want if n not in gpu_cols]
        whole_block_on_host = False
        # the library parses at most 16 columns per call (CSV_MAX_COLS): wide files go in chunks of ascending field indices
        for c0 in range(0, len(gpu_cols), 16):
            chunk = gpu_cols[c0:c0 + 16]
            k = len(chunk)
            kinds = [self._device_type(n) for n in chunk]
            fidx = (ctypes.c_int * k)(*[self._names.index(n) for n in chunk])
            types = (ctypes.c_int * k)(*kinds)
            # a string column is born as the int32 codes of its running dictionary (the one from_arrow() fills on the host route)
            dicts = (ctypes.c_void_p * k)(*[self._dictionary(n).handle() if t == L.CSV_STRING else None for n, t in zip(chunk, kinds)])
            out = (L.DCol * k)()
            n_rows = ctypes.c_int64(0)
            fb = (ctypes.c_int * (k + 2))()
            L.check(lib.vnm_csv_parse_block_ex(text_ptr, length, 0, ord(","), len(self._names), k, fidx, types, dicts, out, ctypes.byref(n_rows), fb, None))
            owned = []
            for i, (n, t) in enumerate(zip(chunk, kinds)):
                if t == L.CSV_STRING:
                    col = _OwnedColumn(out[i], pa.int32())
                    col.dictionary = self._dictionary(n)
                    if not (fb[k] or fb[k + 1] or fb[i]):      # (the column was encoded in this call: its new values join the host copy)
                        col.dictionary.absorb_new()
                elif t == L.CSV_BOOL:
                    col = _OwnedColumn(out[i], pa.int32())
                    col.dictionary = self._dictionary(n)
                else:
                    col = _OwnedColumn(out[i], self.schema.field(n).type)
                owned.append(col)
            if fb[k] or fb[k + 1]:          # a newline inside a quoted value / ragged or empty rows: the whole block goes through pyarrow (which raises on ragged rows)
                whole_block_on_host = True
                break
            assert nrows is None or nrows == n_rows.value
            nrows = n_rows.value
            for i, n in enumerate(chunk):
                if fb[i]:
                    host_cols.append(n)    # a field the exact device parser does not cover
                else:
                    cols[n] = owned[i]
        if whole_block_on_host:
            host_cols, cols, nrows = list(self._want), {}, None
        if host_cols:
            t = self._host_parse(bytes(self._buf[off:off + length]), host_cols)
            # pyarrow's row count is the reference's (it skips empty lines); device-parsed columns of the same block must agree
            if nrows is not None and nrows != t.num_rows:
                raise pa.ArrowInvalid(f"CSV block: the device parser saw {nrows} rows, pyarrow {t.num_rows}")
            nrows = t.num_rows
            b = t.combine_chunks().to_batches()[0] if t.num_rows else None
            staged = DeviceRecordBatch.from_arrow(b, self._dicts) if b is not None else None
            for n in host_cols:
                cols[n] = staged.columns[n] if staged is not None else DeviceColumn.from_arrow(pa.array([], self.schema.field(n).type))
        return DeviceRecordBatch({n: cols[n] for n in self._want}, nrows or 0)

    def read_next_batch(self) -> pa.RecordBatch:
        return self.read_next_device_batch().to_arrow()

    def close(self):
        try:
            self._f.close()
        except Exception:
            pass


def stream_csv(input_file, columns: Optional[Sequence[str]] = None, block_size: int = 64 << 20, numeric_only: bool = False) -> GpuCsvReader:
    """vinum.stream_csv (vinum/io/arrow.py:9-61) with the int64 / float64 / string / date32 / timestamp columns tokenised and parsed
    on the device."""
    return GpuCsvReader(input_file, columns=columns, block_size=block_size, numeric_only=numeric_only)


# ---- Parquet / JSON sources (round 6; vinum/io/arrow.py:111-148 read_json, :151-248 read_parquet) -------------------------------------
# The reference reads both through pyarrow into one host Table.  Decoding stays with pyarrow here too (Parquet pages and JSON text are
# CPU formats); what changes is the way into HBM: record batches of `batch_rows` rows go through the library's PINNED staging ring
# (vnm_stage_column: pinned double buffer, copy threads, one DMA per 32 MB -- the path TableBatchReader uses), numeric columns as they
# are, string / bool / decimal columns as the int32 codes of their running device dictionary.  Row groups are read one at a time: the
# host never holds more than a row group, HBM never more than the batches the consumer keeps.
def _stage_arrow_column(arr, arrow_type) -> DeviceColumn:
    """one host Arrow array of a numeric type -> an HBM column through vnm_stage_column (the pinned ring), buffers owned by the pool"""
    import numpy as np
    from .device import physical_type
    if isinstance(arr, pa.ChunkedArray):
        arr = arr.combine_chunks() if arr.num_chunks != 1 else arr.chunk(0)
    vt, _ = physical_type(arr.type)
    bufs = arr.buffers()
    n = len(arr)
    if n == 0 or bufs[1] is None:
        return DeviceColumn.from_arrow(arr)
    vals = np.frombuffer(bufs[1], dtype=np.uint8)
    validity = np.frombuffer(bufs[0], dtype=np.uint8) if (arr.null_count > 0 and bufs[0] is not None) else None
    out = L.DCol()
    L.check(L.lib().vnm_stage_column(ctypes.c_void_p(vals.ctypes.data), ctypes.c_void_p(validity.ctypes.data) if validity is not None else None,
                                     arr.offset, n, vt, ctypes.byref(out), None))
    L.check(L.lib().vnm_device_synchronize())      # (the host buffers may go once the call returns)
    return _OwnedColumn(out, arrow_type)


class _ArrowBatchSource:
    """record batches from a pyarrow reader -> DeviceRecordBatches staged through the pinned ring"""

    def __init__(self, batches, schema: pa.Schema, columns: Optional[Sequence[str]] = None):
        self._it = iter(batches)
        self.schema = schema if columns is None else pa.schema([schema.field(c) for c in columns])
        self._want = list(self.schema.names)
        self._dicts = {}

    def read_next_device_batch(self) -> DeviceRecordBatch:
        from .device import is_supported
        b = next(self._it)          # StopIteration ends the stream, as in GpuCsvReader
        if isinstance(b, pa.Table):
            b = b.combine_chunks().to_batches()[0] if b.num_rows else pa.RecordBatch.from_pylist([], schema=b.schema)
        cols = {}
        for n in self._want:
            arr = b.column(b.schema.names.index(n))
            if is_supported(arr.type):
                cols[n] = _stage_arrow_column(arr, arr.type)
            else:
                from .vinum_lib import KeyDictionary
                if n not in self._dicts:
                    self._dicts[n] = KeyDictionary(arr.type)
                cols[n] = DeviceColumn.from_arrow(arr, dictionary=self._dicts[n])
        return DeviceRecordBatch(cols, b.num_rows)

    def read_next_batch(self) -> pa.RecordBatch:
        return self.read_next_device_batch().to_arrow()

    def __iter__(self):
        return self

    def __next__(self) -> DeviceRecordBatch:
        return self.read_next_device_batch()


def stream_parquet(source, columns: Optional[Sequence[str]] = None, batch_rows: int = 1 << 24, filters=None, **kwargs) -> _ArrowBatchSource:
    """vinum.read_parquet (vinum/io/arrow.py:151-248) as a STREAM of device record batches: pyarrow decodes one row group at a time
    (`ParquetFile.iter_batches`; `filters` / partitioned datasets through `pyarrow.parquet.read_table`, then sliced), every batch is
    staged into HBM through the pinned ring.  Feed the batches to an operator (`DeviceAggregate.next`, `SortOperator`) like
    stream_csv's."""
    import pyarrow.parquet as pq
    if filters is not None or kwargs:
        t = pq.read_table(source, columns=list(columns) if columns else None, filters=filters, **kwargs)
        return _ArrowBatchSource(t.to_batches(max_chunksize=batch_rows), t.schema, None)
    f = pq.ParquetFile(source)
    schema = f.schema_arrow
    return _ArrowBatchSource(f.iter_batches(batch_size=batch_rows, columns=list(columns) if columns else None), schema, columns)


def read_parquet(source, columns: Optional[Sequence[str]] = None, **kwargs) -> DeviceRecordBatch:
    """the whole file as ONE device record batch (the reference's read_parquet returns one Table)"""
    import pyarrow.parquet as pq
    t = pq.read_table(source, columns=list(columns) if columns else None, **kwargs).combine_chunks()
    return next(_ArrowBatchSource([t], t.schema, None))


def read_json(input_file, read_options=None, parse_options=None, batch_rows: int = 1 << 24) -> _ArrowBatchSource:
    """vinum.read_json (vinum/io/arrow.py:111-148: line-delimited JSON through pyarrow.json) -> device record batches through the pinned ring"""
    import pyarrow.json as pj
    t = pj.read_json(input_file, read_options, parse_options)
    return _ArrowBatchSource(t.to_batches(max_chunksize=batch_rows), t.schema, None)
