"""Shared helpers for the parity tests: golden loading, canonical ordering, exact / ULP comparison."""
import json
import os

import numpy as np
import pyarrow as pa

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def read_ipc(name) -> pa.Table:
    with pa.memory_map(os.path.join(GOLDEN, name), "r") as src:
        return pa.ipc.open_file(src).read_all()


def read_ipc_batches(name):
    with pa.memory_map(os.path.join(GOLDEN, name), "r") as src:
        r = pa.ipc.open_file(src)
        return [r.get_batch(i) for i in range(r.num_record_batches)]


def sliced_batches(table: pa.Table, chunk):
    """Same batching as tests/golden/gen_golden.py::sliced_batches (non-zero Arrow offsets)."""
    t = table.combine_chunks()
    out = []
    for start in range(0, t.num_rows, chunk):
        out.extend(t.slice(start, chunk).to_batches())
    return out


def _bits(arr: pa.Array):
    """(valid: bool ndarray, values: list-or-ndarray comparable bitwise)."""
    if isinstance(arr, pa.ChunkedArray):
        arr = arr.combine_chunks()
    n = len(arr)
    valid = np.ones(n, bool) if arr.null_count == 0 else np.array(arr.is_valid())
    t = arr.type
    if pa.types.is_decimal(t):
        vals = [int(x) if x is not None else 0 for x in arr.to_pylist()]
        return valid, vals
    width = t.bit_width // 8
    view_t = {1: pa.uint8(), 2: pa.uint16(), 4: pa.uint32(), 8: pa.uint64()}[width]
    raw = arr.view(view_t) if t != view_t else arr
    vals = raw.fill_null(0).to_numpy(zero_copy_only=False).astype(np.uint64)
    vals = np.where(valid, vals, 0)
    return valid, vals


def canon(batch, key_names):
    """Order rows canonically by the key columns' (null flag, bit pattern); with no key columns, by all."""
    if isinstance(batch, pa.Table):
        batch = batch.combine_chunks().to_batches()[0] if batch.num_rows else pa.RecordBatch.from_arrays(
            [pa.array([], f.type) for f in batch.schema], names=batch.schema.names)
    if batch.num_rows <= 1:
        return batch
    idxs = [batch.schema.names.index(k) for k in key_names]
    idxs += [i for i in range(batch.num_columns) if i not in idxs]  # tie-break: remaining columns
    keys = []
    for i in idxs:
        valid, vals = _bits(batch.column(i))
        keys.append([(0 if v else 1, int(x)) for v, x in zip(valid, vals)])
    order = sorted(range(batch.num_rows), key=lambda r: tuple(k[r] for k in keys))
    return batch.take(pa.array(order, pa.int64()))


def _ulp_diff(a: np.ndarray, b: np.ndarray):
    """Distance in units-in-the-last-place between two float arrays (same dtype)."""
    it = np.int64 if a.dtype == np.float64 else np.int32
    ai = a.view(it).astype(np.int64)
    bi = b.view(it).astype(np.int64)
    sign = np.int64(np.iinfo(it).min)
    ai = np.where(ai < 0, sign - ai, ai)
    bi = np.where(bi < 0, sign - bi, bi)
    return np.abs(ai - bi)


def assert_col_equal(a: pa.Array, e: pa.Array, name="", ulps=0, check_type=True):
    """Bit-exact for ints / keys; floats bit-exact when ulps == 0 else within `ulps` ULP (NaN == NaN)."""
    if isinstance(a, pa.ChunkedArray):
        a = a.combine_chunks()
    if isinstance(e, pa.ChunkedArray):
        e = e.combine_chunks()
    assert len(a) == len(e), f"{name}: length {len(a)} != {len(e)}"
    if check_type:
        assert a.type == e.type, f"{name}: type {a.type} != {e.type}"
    va, xa = _bits(a)
    ve, xe = _bits(e)
    assert np.array_equal(va, ve), f"{name}: validity differs at rows {np.nonzero(va != ve)[0][:8]}"
    if pa.types.is_decimal(a.type):
        assert xa == xe, f"{name}: decimal values differ"
        return
    if ulps and pa.types.is_floating(a.type):
        dt = np.float64 if pa.types.is_float64(a.type) else np.float32
        fa = a.fill_null(0).to_numpy(zero_copy_only=False).astype(dt)
        fe = e.fill_null(0).to_numpy(zero_copy_only=False).astype(dt)
        both_nan = np.isnan(fa) & np.isnan(fe)
        d = _ulp_diff(np.where(both_nan, 0, fa).astype(dt), np.where(both_nan, 0, fe).astype(dt))
        d = np.where(va, d, 0)
        bad = np.nonzero(d > ulps)[0]
        assert bad.size == 0, f"{name}: {bad.size} rows differ by > {ulps} ULP, e.g. row {bad[0]}: {fa[bad[0]]!r} vs {fe[bad[0]]!r}"
        return
    bad = np.nonzero(np.asarray(xa) != np.asarray(xe))[0]
    assert bad.size == 0, (f"{name}: {bad.size} rows differ bitwise, e.g. row {bad[0]}: "
                           f"{a[int(bad[0])]} vs {e[int(bad[0])]}")


def assert_batches_equal(actual, expected, key_names=None, float_ulps=0, positional=False, what=""):
    """Canonicalise both by key columns and compare column by column."""
    if key_names is not None:
        actual = canon(actual, key_names)
        expected = canon(expected, key_names)
    assert actual.num_columns == expected.num_columns, f"{what}: column count"
    assert actual.num_rows == expected.num_rows, f"{what}: rows {actual.num_rows} != {expected.num_rows}"
    if not positional:
        assert actual.schema.names == expected.schema.names, f"{what}: {actual.schema.names} != {expected.schema.names}"
    for i in range(actual.num_columns):
        assert_col_equal(actual.column(i), expected.column(i), f"{what}:{actual.schema.names[i]}", ulps=float_ulps)


def assert_agg_equal(actual, expected, funcs, key_names, exact_float_inputs=("v_f64q", "fare", "v"), what=""):
    """Aggregate parity: bit-exact for keys, counts, integer sums, decimals, MIN/MAX and float sums over
    exactly-representable (quantised) inputs; SUM/AVG over arbitrary floats are order dependent (the
    reference adds in row order, agg_funcs.h:294-305) so those columns use rtol=1e-12, atol=1e-9
    (the reference's own tests use np.allclose defaults, rtol=1e-5: vinum/tests/conftest.py:128-142)."""
    actual = canon(actual, key_names)
    expected = canon(expected, key_names)
    assert actual.num_rows == expected.num_rows, f"{what}: rows {actual.num_rows} != {expected.num_rows}"
    assert actual.schema.names == expected.schema.names, f"{what}: {actual.schema.names} != {expected.schema.names}"
    loose = set()
    for f, col, out in funcs:
        if f in (4, 5) and col and col not in exact_float_inputs:
            loose.add(out)
    for i, name in enumerate(actual.schema.names):
        a, e = actual.column(i), expected.column(i)
        if name in loose and pa.types.is_floating(e.type):
            assert a.type == e.type, f"{what}:{name}: type {a.type} != {e.type}"
            va, _ = _bits(a)
            ve, _ = _bits(e)
            assert np.array_equal(va, ve), f"{what}:{name}: validity differs"
            fa = a.fill_null(0).to_numpy(zero_copy_only=False).astype(np.float64)
            fe = e.fill_null(0).to_numpy(zero_copy_only=False).astype(np.float64)
            ok = np.isclose(fa, fe, rtol=1e-12, atol=1e-9, equal_nan=True)
            assert ok.all(), f"{what}:{name}: {(~ok).sum()} rows beyond rtol=1e-12/atol=1e-9, e.g. {fa[~ok][0]!r} vs {fe[~ok][0]!r}"
        else:
            assert_col_equal(a, e, f"{what}:{name}")
