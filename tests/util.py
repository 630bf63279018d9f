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
    if pa.types.is_string(t) or pa.types.is_large_string(t) or pa.types.is_binary(t) or pa.types.is_large_binary(t):
        return valid, [x if x is not None else "" for x in arr.to_pylist()]     # (string MIN / MAX, string group keys)
    if pa.types.is_boolean(t):
        return valid, np.where(valid, arr.fill_null(False).to_numpy(zero_copy_only=False).astype(np.uint64), 0)
    width = t.bit_width // 8
    view_t = {1: pa.uint8(), 2: pa.uint16(), 4: pa.uint32(), 8: pa.uint64()}[width]
    raw = arr.view(view_t) if t != view_t else arr
    vals = raw.fill_null(0).to_numpy(zero_copy_only=False).astype(np.uint64)
    vals = np.where(valid, vals, 0)
    if pa.types.is_floating(t):
        # NaN sign / payload bits are not part of the comparison: the reference's own NaN bits depend on the CPU it runs on
        # (x86 SUBSD hands back the NaN operand, 0/0 is the NEGATIVE "real indefinite"; gfx950 negates the NaN of `a - b`
        # and generates positive NaNs).  Every NaN compares as the canonical quiet NaN.
        f = arr.fill_null(0).to_numpy(zero_copy_only=False)
        canon_nan = {2: 0x7E00, 4: 0x7FC00000, 8: 0x7FF8000000000000}[width]
        vals = np.where(np.isnan(f) & valid, np.uint64(canon_nan), vals)
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
    cols = [_bits(batch.column(i)) for i in idxs]
    if all(isinstance(vals, np.ndarray) for _, vals in cols):
        # (null flag, bit pattern) per column, first column most significant: one lexsort instead of a Python sort of tuples
        # (results with ~1e6 groups made the large dense-path tests spend most of their time here)
        sort_keys = []
        for valid, vals in cols:
            sort_keys.append((~valid).astype(np.uint8))
            sort_keys.append(vals.astype(np.uint64))
        order = np.lexsort(sort_keys[::-1])
        return batch.take(pa.array(order.astype(np.int64)))
    keys = [[(0 if v else 1, int(x)) for v, x in zip(valid, vals)] for valid, vals in cols]
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
    if isinstance(xa, list):
        assert xa == xe, f"{name}: values differ: {[(x, y) for x, y in zip(xa, xe) if x != y][:4]}"
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


def _schema_of(source):
    return source.schema if isinstance(source, pa.Table) else list(source)[0].schema


def exact_group_sums(source, key_names, col):
    """{canonical key tuple: (math.fsum of the non-NULL values of `col`, their count)} over the record batches `source`
    (the rows that were actually aggregated, i.e. after the predicate).  float32 inputs are summed as the doubles they convert
    to, as the reference and the device do (agg_func_factory.cpp:126-131)."""
    import math
    t = pa.Table.from_batches(list(source)).combine_chunks() if not isinstance(source, pa.Table) else source.combine_chunks()
    n = t.num_rows
    keycols = []
    for k in key_names:
        valid, vals = _bits(t.column(k))
        keycols.append((~valid).astype(np.uint64))
        keycols.append(np.asarray(vals, dtype=np.uint64))
    arr = t.column(col).combine_chunks()
    vvalid = np.ones(n, bool) if arr.null_count == 0 else np.array(arr.is_valid())
    x = arr.fill_null(0).to_numpy(zero_copy_only=False).astype(np.float64)
    if keycols:
        stacked = np.stack(keycols, axis=1) if n else np.zeros((0, len(keycols)), np.uint64)
        uniq, inv = np.unique(stacked, axis=0, return_inverse=True)
        inv = np.asarray(inv).reshape(-1)
    else:
        uniq, inv = np.zeros((1, 0), np.uint64), np.zeros(n, np.int64)
    order = np.argsort(inv, kind="stable")
    inv_s, x_s, v_s = inv[order], x[order], vvalid[order]
    starts = np.searchsorted(inv_s, np.arange(len(uniq)), side="left")
    ends = np.searchsorted(inv_s, np.arange(len(uniq)), side="right")
    out = {}
    for g in range(len(uniq)):
        seg = x_s[starts[g]:ends[g]][v_s[starts[g]:ends[g]]]
        try:
            exact = math.fsum(seg.tolist())
        except (ValueError, OverflowError):     # inf - inf, or an intermediate overflow: not a finite exact value (the caller compares
            exact = float(np.sum(seg))          # the class of the result -- NaN / +-inf -- with the reference's)
        out[tuple(int(w) for w in uniq[g])] = (exact, int(seg.size))
    return out


def _assert_float_agg_exact(a, e, f, exact_rows, name, what):
    """SUM / AVG over arbitrary floats.  The reference adds sequentially in row order (agg_funcs.h:294-305), so its result depends
    on the order; ours is the correctly rounded exact sum whatever the order.  Asserted: ours is within 1 ULP of the exact value
    (math.fsum; AVG: fsum / count, the division rounds once more) and never further from the reference than the reference is from
    the exact value (+ 1 ULP) -- the bound of tests/test_gpu_float.py, instead of an rtol."""
    assert a.type == e.type, f"{what}:{name}: type {a.type} != {e.type}"
    va, _ = _bits(a)
    ve, _ = _bits(e)
    assert np.array_equal(va, ve), f"{what}:{name}: validity differs"
    dt = np.float64 if pa.types.is_float64(e.type) else np.float32
    fa = a.fill_null(0).to_numpy(zero_copy_only=False).astype(dt)
    fe = e.fill_null(0).to_numpy(zero_copy_only=False).astype(dt)
    ex = np.array([(s if f == 4 else (s / c if c else 0.0)) for s, c in exact_rows], dtype=np.float64).astype(dt)
    live = va & np.isfinite(ex.astype(np.float64)) & np.isfinite(fe.astype(np.float64))
    d_ours = np.where(live, _ulp_diff(np.where(live, fa, 0).astype(dt), np.where(live, ex, 0).astype(dt)), 0)
    bad = np.nonzero(d_ours > 1)[0]
    assert bad.size == 0, f"{what}:{name}: {bad.size} groups are not the (correctly rounded) exact value, e.g. row {bad[0]}: {fa[bad[0]]!r} vs exact {ex[bad[0]]!r}"
    d_ref = np.where(live, _ulp_diff(np.where(live, fe, 0).astype(dt), np.where(live, ex, 0).astype(dt)), 0)
    d_both = np.where(live, _ulp_diff(np.where(live, fa, 0).astype(dt), np.where(live, fe, 0).astype(dt)), 0)
    bad = np.nonzero(d_both > d_ref + 1)[0]
    assert bad.size == 0, f"{what}:{name}: {bad.size} groups further from the reference than the reference is from the exact value"
    # non-finite results (inf / NaN inputs, overflow): the same class of value as the reference
    nf = va & ~live
    assert np.array_equal(np.isnan(fa[nf]), np.isnan(fe[nf])) and np.array_equal(fa[nf][~np.isnan(fa[nf])], fe[nf][~np.isnan(fe[nf])]), \
        f"{what}:{name}: non-finite results differ"


def assert_agg_equal(actual, expected, funcs, key_names, exact_float_inputs=("v_f64q", "fare", "v"), what="", source=None):
    """Aggregate parity: bit-exact for keys, counts, integer sums, decimals, MIN/MAX and float sums over
    exactly-representable (quantised) inputs.  SUM/AVG over arbitrary floats are order dependent in the reference (it adds in
    row order, agg_funcs.h:294-305): with `source` (the record batches that were aggregated, after any predicate) those columns
    are held to the exact-sum bound of _assert_float_agg_exact; without it they must agree BIT FOR BIT (true on quantised inputs; the
    reference's own tests use np.allclose defaults, rtol=1e-5: vinum/tests/conftest.py:128-142)."""
    actual = canon(actual, key_names)
    expected = canon(expected, key_names)
    assert actual.num_rows == expected.num_rows, f"{what}: rows {actual.num_rows} != {expected.num_rows}"
    assert actual.schema.names == expected.schema.names, f"{what}: {actual.schema.names} != {expected.schema.names}"
    loose = {}
    for f, col, out in funcs:
        if f in (4, 5) and col and col not in exact_float_inputs:
            loose[out] = (f, col)
    exact_cache = {}
    for i, name in enumerate(actual.schema.names):
        a, e = actual.column(i), expected.column(i)
        if name in loose and source is not None and not pa.types.is_floating(_schema_of(source).field(loose[name][1]).type):
            assert_col_equal(a, e, f"{what}:{name}")     # SUM / AVG of integers and temporals: exact arithmetic, bit for bit
        elif name in loose and pa.types.is_floating(e.type) and source is not None and all(k in expected.schema.names for k in key_names):
            f, col = loose[name]
            if col not in exact_cache:
                exact_cache[col] = exact_group_sums(source, key_names, col)
            kb = [_bits(expected.column(k)) for k in key_names]
            rows = []
            for r in range(expected.num_rows):
                key = tuple(x for valid, vals in kb for x in (int(not valid[r]), int(vals[r])))
                rows.append(exact_cache[col][key])
            _assert_float_agg_exact(a, e, f, rows, name, what)
        elif name in loose and pa.types.is_floating(e.type):
            # no `source`: there is no exact value to hold an order-dependent float sum to, and an rtol is not a parity bar
            #  -- such a column has to agree bit for bit (it does on quantised inputs, where every partial sum is
            # exact); anything else must pass the aggregated batches as `source=`
            try:
                assert_col_equal(a, e, f"{what}:{name}")
            except AssertionError as err:
                raise AssertionError(f"{what}:{name}: float SUM / AVG of '{loose[name][1]}' differs bitwise from the reference and no "
                                     f"`source=` was given for the exact-sum bound ({err})") from None
        else:
            assert_col_equal(a, e, f"{what}:{name}")


def random_agg_case(rng, specials=False):
    """Random aggregate query for the differential tests: (columns dict, key names, input names, funcs, n, groups,
    skew).  1-3 key columns of mixed widths (some with NULLs), 0-3 typed input columns (some with NULLs).
    specials: float inputs carry NaNs, -0.0 and +0.0 (everywhere, or only in the second half of the rows) and are read by
    every function -- the row-order dependent part of MinMaxFunc::Update (agg_funcs.h:188-201), the sign of an all-(-0.0) SUM."""
    from oracle import oracle as O
    n = int(rng.integers(150_000, 420_000))
    groups = int(rng.choice([3, 40, 900, 5_000, 60_000, 250_000]))
    nkeys = int(rng.choice([1, 1, 1, 2, 3]))
    skew = rng.random() < 0.3

    def int_col(lo, hi, dtype, null_p):
        a = rng.integers(lo, hi, n).astype(dtype)
        return pa.array(a, mask=(rng.random(n) < null_p) if null_p else None)

    cols = {}
    per_key = max(2, int(round(groups ** (1.0 / nkeys))))
    for j in range(nkeys):
        u = rng.random(n)
        if skew:
            u = u ** 6
        vals = np.floor(u * per_key).astype(np.int64)
        kind = rng.choice(["i64", "i32", "f64", "u8"]) if (nkeys > 1 or rng.random() < 0.3) else "i64"
        null_p = 0.05 if rng.random() < 0.3 else 0.0
        mask = (rng.random(n) < null_p) if null_p else None
        if kind == "i64":
            arr = pa.array(vals * 7919 - 13, mask=mask)
        elif kind == "i32":
            arr = pa.array((vals - per_key // 2).astype(np.int32), mask=mask)
        elif kind == "u8":
            arr = pa.array((vals % 251).astype(np.uint8), mask=mask)
        else:
            arr = pa.array(vals.astype(np.float64) * 0.5 - 1.0, mask=mask)
        cols[f"k{j}"] = arr
    key_names = list(cols)
    makers = {
        "f64": lambda p: pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0, mask=(rng.random(n) < p) if p else None),
        "i64": lambda p: int_col(-2**45, 2**45, np.int64, p),
        "i32": lambda p: int_col(-2**31, 2**31 - 1, np.int32, p),
        "u16": lambda p: int_col(0, 2**16, np.uint16, p),
        "u64": lambda p: pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2), mask=(rng.random(n) < p) if p else None),
        "f32": lambda p: pa.array((rng.integers(0, 2**10, n) / 8.0).astype(np.float32), mask=(rng.random(n) < p) if p else None),
    }
    ninputs = int(rng.choice([0, 1, 1, 2, 3]))
    in_names = []
    for c in range(ninputs):
        t = str(rng.choice(list(makers)))
        cols[f"v{c}"] = makers[t](0.15 if rng.random() < 0.35 else 0.0)
        in_names.append(f"v{c}")
    funcs = [(O.COUNT_STAR, "", "n")] if (ninputs == 0 or rng.random() < 0.5) else []
    for name in in_names:
        if specials and pa.types.is_floating(cols[name].type):
            a = cols[name]
            vals = a.fill_null(0).to_numpy(zero_copy_only=False).copy()
            u = rng.random(n)
            first = 0 if rng.random() < 0.5 else n // 2
            u[:first] = 1.0
            vals[u < 0.01] = np.nan
            vals[(u >= 0.01) & (u < 0.03)] = -0.0
            vals[(u >= 0.03) & (u < 0.06)] = 0.0
            if rng.random() < 0.4:
                vals = np.where(np.isnan(vals), vals, -vals)      # zero is the MAXIMUM of many groups
            cols[name] = pa.array(vals, mask=None if a.null_count == 0 else ~np.array(a.is_valid()))
            # (SUM / AVG too: the inputs are quantised, so the sums are exact in any order -- a group whose inputs are all -0.0 sums
            # to -0.0, SumFunc starts from the first value: agg_funcs.h:286-305)
        picks = rng.choice([O.SUM, O.AVG, O.MIN, O.MAX, O.COUNT], size=int(rng.integers(1, 4)), replace=False)
        for f in picks:
            funcs.append((int(f), name, f"f{len(funcs)}"))
    return cols, key_names, in_names, funcs, n, groups, skew
