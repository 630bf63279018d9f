"""Host logic of libvinum_hip.so without a GPU: the lowering of (function, input type) onto 64-bit accumulator
words (vnm_agg_plan_host) and the finalisation of result columns from those words (vnm_agg_finalize_host:
AVG incl. the 128-bit divmod path, int64 SUM -> decimal128 promotion, MIN/MAX decoding, int32 wrap of time32).
The accumulate step is emulated in NumPy exactly as the kernels do it (op kinds of include/vinum_hip.h), then the
finalised columns must equal the oracle bit for bit."""
import ctypes

import numpy as np
import pyarrow as pa
import pytest

from oracle import oracle as O
from tests import util
from vinum_amd import _lib as L
from vinum_amd.device import physical_type
from vinum_amd.ops import _func_array


def _ints(xs):
    return (ctypes.c_int * max(len(xs), 1))(*xs)


def _enc_i64(x):
    return (x.astype(np.int64).view(np.uint64)) ^ np.uint64(1 << 63)


def _enc_f64(d):
    b = d.astype(np.float64).view(np.uint64)
    neg = (b >> np.uint64(63)).astype(bool)
    return np.where(neg, ~b, b | np.uint64(1 << 63))


def _emulate(table, group_col, funcs):
    lib = L.load()
    schema = table.schema
    ft = [f for f, _, _ in funcs]
    it, fl, ids = [], [], []
    for f, col, _ in funcs:
        if col:
            p, g = physical_type(schema.field(col).type)
            it.append(p); fl.append(g); ids.append(schema.get_field_index(col))
        else:
            it.append(L.U64); fl.append(0); ids.append(-1)
    kt = [physical_type(schema.field(group_col).type)[0]]
    nkw, nw, nops = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    mk = (ctypes.c_int * 40)()
    ops = (ctypes.c_int * (3 * 48))()
    L.check(lib.vnm_agg_plan_host(L.SINGLE_NUMERICAL, 1, _ints(kt), len(ft), _ints(ft), _ints(it), _ints(fl), _ints(ids),
                                  ctypes.byref(nkw), ctypes.byref(nw), mk, ctypes.byref(nops), ops))
    # distinct columns in order of first use (same rule as the library)
    dcols = []
    for f, col, _ in funcs:
        if col and col not in dcols:
            dcols.append(col)
    keys = table.column(group_col).to_numpy()
    uk, inv = np.unique(keys, return_inverse=True)
    G = len(uk)
    words = np.zeros((nw.value, G), np.uint64)
    for w in range(nw.value):
        if mk[w] == 2:
            words[w, :] = np.uint64(0xFFFFFFFFFFFFFFFF)
    for o in range(nops.value):
        kind, col, w = ops[3 * o], ops[3 * o + 1], ops[3 * o + 2]
        if kind == 0:
            np.add.at(words[w], inv, np.uint64(1)); continue
        arr = table.column(dcols[col]).combine_chunks()
        valid = np.ones(len(arr), bool) if arr.null_count == 0 else np.array(arr.is_valid())
        pt, _ = physical_type(arr.type)
        raw = arr.view({1: pa.int8(), 2: pa.int16(), 4: pa.int32(), 8: pa.int64()}[arr.type.bit_width // 8]) \
            if not pa.types.is_floating(arr.type) and not pa.types.is_integer(arr.type) else arr
        vals = raw.fill_null(0).to_numpy(zero_copy_only=False)
        iv, vv = inv[valid], vals[valid]
        if kind == 1:
            np.add.at(words[w], iv, np.uint64(1))
        elif kind == 2:
            acc = words[w].view(np.float64)
            np.add.at(acc, iv, vv.astype(np.float64))
        elif kind == 3:
            np.add.at(words[w], iv, vv.astype(np.int64).view(np.uint64) if pt not in (L.U8, L.U16, L.U32, L.U64) else vv.astype(np.uint64))
        elif kind == 4:
            np.add.at(words[w], iv, vv.astype(np.int64 if pt == L.I64 else np.uint64).view(np.uint64) & np.uint64(0xFFFFFFFF))
        elif kind == 5:
            np.add.at(words[w], iv, (vv.astype(np.int64) >> 32).view(np.uint64))
        elif kind == 6:
            np.add.at(words[w], iv, vv.astype(np.uint64) >> np.uint64(32))
        else:
            if pt in (L.F32, L.F64): e = _enc_f64(vv)
            elif pt in (L.U8, L.U16, L.U32, L.U64): e = vv.astype(np.uint64)
            else: e = _enc_i64(vv)
            (np.minimum if kind == 7 else np.maximum).at(words[w], iv, e)
    # finalise every function on the host
    arrays = []
    wp = (ctypes.c_void_p * nw.value)(*[words[w].ctypes.data for w in range(nw.value)])
    for i, (f, col, out) in enumerate(funcs):
        cells = np.zeros((G, 2), np.uint64); valid = np.zeros(G, np.uint8); kind = ctypes.c_int()
        L.check(lib.vnm_agg_finalize_host(L.SINGLE_NUMERICAL, 1, _ints(kt), len(ft), _ints(ft), _ints(it), _ints(fl), _ints(ids),
                                          i, G, wp, cells.ctypes.data, valid.ctypes.data, ctypes.byref(kind)))
        arrays.append(_func_array(f, schema.field(col).type if col else None, kind.value, cells, valid))
    return pa.RecordBatch.from_arrays([pa.array(uk)] + arrays, names=[group_col] + [o for _, _, o in funcs])


@pytest.mark.parametrize("seed", [0, 1])
def test_plan_and_finalize_match_oracle(seed):
    rng = np.random.default_rng(seed)
    n = 5000
    t = pa.table({
        "k": rng.integers(0, 40, n).astype(np.int64),
        "i8": pa.array(rng.integers(-128, 128, n).astype(np.int8), mask=rng.random(n) < 0.1),
        "u16": pa.array(rng.integers(0, 65536, n).astype(np.uint16), mask=rng.random(n) < 0.1),
        "i64": pa.array(rng.integers(-2**62, 2**62, n).astype(np.int64), mask=rng.random(n) < 0.1),
        "big": pa.array((2**63 - 1 - rng.integers(0, 100, n)).astype(np.int64), mask=rng.random(n) < 0.1),
        "u64": pa.array((np.uint64(2**64 - 1) - rng.integers(0, 100, n).astype(np.uint64)), mask=rng.random(n) < 0.1),
        "f64": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0, mask=rng.random(n) < 0.1),
        "f32": pa.array((rng.integers(-2**10, 2**10, n) / 8).astype(np.float32), mask=rng.random(n) < 0.1),
        "t32": pa.array(rng.integers(0, 2**31 - 1, n).astype(np.int32), mask=rng.random(n) < 0.1).view(pa.time32("s")),
        "sparse": pa.array(rng.normal(size=n), mask=np.ones(n, bool)),   # all NULL -> every result NULL
    })
    funcs = [(O.COUNT_STAR, "", "n")]
    for col in ["i8", "u16", "i64", "big", "u64", "f64", "f32", "t32", "sparse"]:
        for f, nm in [(O.COUNT, "cnt"), (O.MIN, "min"), (O.MAX, "max"), (O.SUM, "sum"), (O.AVG, "avg")]:
            if len(funcs) < 46:
                funcs.append((f, col, f"{nm}_{col}"))
    for chunk in (funcs[:16], funcs[16:31], funcs[31:46]):
        got = _emulate(t, "k", chunk)
        o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], chunk)
        for b in t.to_batches():
            o.next(b)
        util.assert_batches_equal(got, o.result(), key_names=["k"], what="host finalize")


def test_plan_errors_are_the_reference_messages():
    lib = L.load()
    nk, nw, nops = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = lib.vnm_agg_plan_host(L.SINGLE_NUMERICAL, 1, _ints([L.I64]), 1, _ints([L.SUM]), _ints([99]), _ints([0]), None,
                               ctypes.byref(nk), ctypes.byref(nw), None, ctypes.byref(nops), None)
    assert rc != 0 and b"not supported by sum()" in lib.vnm_last_error()
    rc = lib.vnm_agg_plan_host(L.ONE_GROUP, 1, _ints([L.I64]), 0, _ints([]), _ints([]), _ints([]), None,
                               ctypes.byref(nk), ctypes.byref(nw), None, ctypes.byref(nops), None)
    assert rc != 0
