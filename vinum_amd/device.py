"""HBM-resident Arrow columns: the device counterpart of vinum/arrow/record_batch.py's pa.Array handling.

A DeviceColumn keeps the Arrow layout (values buffer + optional validity bitmap + offset) in HBM, so a
column staged once can flow through filter -> project -> aggregate -> sort without returning to the
host.  Memory comes from the library's caching allocator (vnm_malloc) or wraps a torch tensor.
"""
import ctypes

import numpy as np
import pyarrow as pa

from . import _lib as L

_NP = {L.I8: np.int8, L.I16: np.int16, L.I32: np.int32, L.I64: np.int64, L.U8: np.uint8, L.U16: np.uint16,
       L.U32: np.uint32, L.U64: np.uint64, L.F32: np.float32, L.F64: np.float64}
_WIDTH = {L.I8: 1, L.U8: 1, L.I16: 2, L.U16: 2, L.I32: 4, L.U32: 4, L.F32: 4, L.I64: 8, L.U64: 8, L.F64: 8}
_FROM_NP = {np.dtype(v): k for k, v in _NP.items()}


def physical_type(t: pa.DataType):
    """Arrow type -> (vnm_type, flags).  Temporal types map to their storage integers
    (the reference's NumericArrayIter<T> does the same, array_iterators.cpp:7-47)."""
    T = pa.types
    if T.is_int8(t): return L.I8, 0
    if T.is_int16(t): return L.I16, 0
    if T.is_int32(t) or T.is_date32(t): return L.I32, 0
    if T.is_time32(t): return L.I32, L.FLAG_SUM32
    if T.is_int64(t) or T.is_date64(t) or T.is_time64(t) or T.is_timestamp(t) or T.is_duration(t): return L.I64, 0
    if T.is_uint8(t): return L.U8, 0
    if T.is_uint16(t): return L.U16, 0
    if T.is_uint32(t): return L.U32, 0
    if T.is_uint64(t): return L.U64, 0
    if T.is_float32(t): return L.F32, 0
    if T.is_float64(t): return L.F64, 0
    raise RuntimeError(f"Unsupported data type for a GPU column: {t}")


def is_supported(t: pa.DataType) -> bool:
    try:
        physical_type(t)
        return True
    except RuntimeError:
        return False


def pool_trim() -> int:
    """Give the HBM the library's caching allocator holds but nobody uses back to the device; returns the bytes released
    (vnm_pool_trim).  Live buffers are untouched."""
    return int(L.lib().vnm_pool_trim())


class DeviceBuffer:
    """Owning handle on library-allocated HBM."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        self.ptr = L.lib().vnm_malloc(max(self.nbytes, 1))
        if not self.ptr:
            raise L.VinumHipError(L.last_error())

    @staticmethod
    def adopt(ptr: int, nbytes: int) -> "DeviceBuffer":
        """Take ownership of a block the library allocated for the caller (vnm_agg_result_device_alloc)."""
        b = DeviceBuffer.__new__(DeviceBuffer)
        b.nbytes = int(nbytes)
        b.ptr = int(ptr)
        return b

    @staticmethod
    def from_host(arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        b = DeviceBuffer(arr.nbytes)
        if arr.nbytes:
            L.check(L.lib().vnm_memcpy_h2d(b.ptr, arr.ctypes.data, arr.nbytes))
        return b

    def to_host(self, dtype, count) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            L.check(L.lib().vnm_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            L.lib().vnm_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceColumn:
    def __init__(self, values, validity, offset, length, arrow_type, keep=None, dictionary=None):
        """values / validity: DeviceBuffer, raw int pointer, or None (validity).
        dictionary: a vinum_lib.KeyDictionary when the column holds the int32 CODES of a non-numeric column (strings,
        bools, decimals: dictionary-encoded on ingest); to_arrow() then returns the decoded values."""
        self._values, self._validity = values, validity
        self.offset, self.length, self.arrow_type = int(offset), int(length), arrow_type
        self.vnm_type, self.flags = physical_type(arrow_type)
        self._keep = keep  # anything that must outlive the raw pointers (torch tensors)
        self.dictionary = dictionary

    def like(self, values, validity, offset, length) -> "DeviceColumn":
        """A column of the same logical type over other buffers (filter / take / slice outputs keep the dictionary)."""
        return DeviceColumn(values, validity, offset, length, self.arrow_type, dictionary=self.dictionary)

    # -- constructors -------------------------------------------------------------------------------
    @staticmethod
    def from_arrow(arr, dictionary=None) -> "DeviceColumn":
        if isinstance(arr, pa.ChunkedArray):
            arr = arr.combine_chunks() if arr.num_chunks != 1 else arr.chunk(0)
        if dictionary is not None:                    # non-numeric column: stage its int32 codes
            col = DeviceColumn.from_arrow(dictionary.encode(arr))
            col.dictionary = dictionary
            return col
        vt, _ = physical_type(arr.type)
        w = _WIDTH[vt]
        bufs = arr.buffers()
        n, off = len(arr), arr.offset
        host_vals = np.frombuffer(bufs[1], dtype=np.uint8)[off * w:(off + n) * w] if (bufs[1] is not None and n) \
            else np.zeros(0, np.uint8)
        vbuf = None
        dev_off = 0
        if arr.null_count > 0 and bufs[0] is not None:
            first, last = off >> 3, (off + n + 7) >> 3
            vbuf = DeviceBuffer.from_host(np.frombuffer(bufs[0], dtype=np.uint8)[first:last])
            dev_off = off & 7
        if dev_off:
            pad = np.zeros(dev_off * w, np.uint8)
            host_vals = np.concatenate([pad, host_vals])
        return DeviceColumn(DeviceBuffer.from_host(host_vals), vbuf, dev_off, n, arr.type)

    @staticmethod
    def from_numpy(a: np.ndarray, arrow_type=None) -> "DeviceColumn":
        a = np.ascontiguousarray(a)
        t = arrow_type or pa.from_numpy_dtype(a.dtype)
        return DeviceColumn(DeviceBuffer.from_host(a.view(np.uint8)), None, 0, len(a), t)

    @staticmethod
    def from_torch(t, arrow_type=None, validity=None) -> "DeviceColumn":
        """Zero-copy view of a contiguous CUDA/HIP torch tensor (bench / fused pipelines)."""
        import torch
        assert t.is_cuda and t.is_contiguous()
        npdt = {torch.float64: np.float64, torch.float32: np.float32, torch.int64: np.int64, torch.int32: np.int32,
                torch.int16: np.int16, torch.int8: np.int8, torch.uint8: np.uint8}[t.dtype]
        at = arrow_type or pa.from_numpy_dtype(np.dtype(npdt))
        vptr = validity.data_ptr() if validity is not None else None
        return DeviceColumn(t.data_ptr(), vptr, 0, t.numel(), at, keep=(t, validity))

    @staticmethod
    def empty(length, arrow_type, with_valid_bytes=False):
        vt, _ = physical_type(arrow_type)
        return DeviceColumn(DeviceBuffer(length * _WIDTH[vt]), None, 0, length, arrow_type)

    # -- views --------------------------------------------------------------------------------------
    @staticmethod
    def _ptr(x):
        if x is None:
            return None
        return x.ptr if isinstance(x, DeviceBuffer) else int(x)

    @property
    def values_ptr(self):
        return self._ptr(self._values)

    @property
    def validity_ptr(self):
        return self._ptr(self._validity)

    def dcol(self) -> L.DCol:
        d = L.DCol()
        d.values = self.values_ptr
        d.validity = self.validity_ptr
        d.offset = self.offset
        d.length = self.length
        d.type = self.vnm_type
        d.flags = self.flags
        return d

    def __len__(self):
        return self.length

    def slice(self, offset: int, length: int) -> "DeviceColumn":
        """View of rows [offset, offset + length): same HBM buffers, Arrow offset moved (no copy)."""
        offset = max(0, min(int(offset), self.length))
        length = max(0, min(int(length), self.length - offset))
        return DeviceColumn(self._values, self._validity, self.offset + offset, length, self.arrow_type, keep=(self, self._keep),
                            dictionary=self.dictionary)

    def to_numpy(self) -> np.ndarray:
        w = _WIDTH[self.vnm_type]
        raw = np.empty((self.offset + self.length) * w, np.uint8)
        if raw.nbytes:
            L.check(L.lib().vnm_memcpy_d2h(raw.ctypes.data, self.values_ptr, raw.nbytes))
        return raw[self.offset * w:].view(_NP[self.vnm_type])

    def to_arrow(self) -> pa.Array:
        vals = self.to_numpy()
        mask = None
        if self.validity_ptr:
            nb = (self.offset + self.length + 7) >> 3
            bits = np.empty(nb, np.uint8)
            L.check(L.lib().vnm_memcpy_d2h(bits.ctypes.data, self.validity_ptr, nb))
            valid = np.unpackbits(bits, bitorder="little")[self.offset:self.offset + self.length].astype(bool)
            mask = ~valid
        out = arrow_from_numpy(vals, mask, self.arrow_type)
        return self.dictionary.decode(out) if self.dictionary is not None else out


def arrow_from_numpy(vals: np.ndarray, mask, t: pa.DataType) -> pa.Array:
    a = pa.array(vals, mask=mask if (mask is not None and mask.any()) else None)
    if a.type != t:
        a = a.view(t)
    return a


def dcol_array(cols):
    """ctypes array of vnm_dcol from DeviceColumns (a None entry becomes a zeroed struct)."""
    arr = (L.DCol * max(len(cols), 1))()
    for i, c in enumerate(cols):
        if c is not None:
            arr[i] = c.dcol()
    return arr
