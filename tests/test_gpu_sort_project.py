"""GPU parity: sort / top-K / take and fused projection (HIP, through the C ABI) vs golden vectors from the
real reference (Sort) and from NumPy ufuncs exactly as the reference dispatches them (projection)."""
import numpy as np
import pyarrow as pa
import pytest
import warnings

from tests import util

pytestmark = pytest.mark.gpu
MAN = util.manifest()


def gpu_sort(table: pa.Table, cols, orders, limit=0) -> pa.RecordBatch:
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    t = table.combine_chunks()
    dev = {n: DeviceColumn.from_arrow(t.column(n)) for n in t.schema.names}
    n = limit if limit else t.num_rows
    sorted_key = None
    if limit:
        idx = ops.sort_indices([dev[c] for c in cols], orders, limit=limit)
    else:
        # the full sort may hand its first key back already sorted (vnm_sort_indices_keyed): it must then be what the gather gives
        idx, sorted_key = ops.sort_indices_keyed([dev[c] for c in cols], orders)
        if sorted_key is not None:
            util.assert_col_equal(sorted_key.to_arrow(), ops.take(dev[cols[0]], idx, n).to_arrow(), f"sorted key {cols[0]} vs gather")
    return pa.RecordBatch.from_arrays([(sorted_key if (sorted_key is not None and name == cols[0]) else ops.take(dev[name], idx, n)).to_arrow()
                                       for name in t.schema.names], names=t.schema.names)


@pytest.mark.parametrize("case", MAN["sort"], ids=lambda c: c["name"])
def test_sort_matches_reference_golden(case):
    table = util.read_ipc(case["input"])
    expected = util.read_ipc(case["expected"])
    got = gpu_sort(table, case["cols"], case["orders"])
    util.assert_batches_equal(got, expected, what=case["name"])  # order-sensitive, bit-exact


@pytest.mark.parametrize("n,k", [(70_000, 10), (300_000, 1000), (2_000_000, 10), (2_000_000, 50_000)])
@pytest.mark.parametrize("desc", [0, 1])
@pytest.mark.parametrize("special", [False, True])
def test_topk_equals_full_sort_prefix(n, k, desc, special):
    """ORDER BY v [DESC] LIMIT K must return exactly the first K rows of the stable full sort
    (the reference sorts everything and slices afterwards, planner.py:478-501)."""
    from oracle import oracle as O
    rng = np.random.default_rng(n + k + desc)
    v = np.round(rng.normal(11, 9, n), 1 if special else 6)   # 1 decimal -> many ties
    mask = None
    if special:
        v[rng.random(n) < 0.001] = np.nan
        mask = rng.random(n) < 0.001
    t = pa.table({"rowid": pa.array(np.arange(n, dtype=np.int64)), "v": pa.array(v, mask=mask)})
    got = gpu_sort(t, ["v"], [desc], limit=k)
    s = O.OracleSort(["v"], [desc])
    for b in t.to_batches():
        s.next(b)
    exp = s.sorted().slice(0, k)
    util.assert_batches_equal(got, exp, what=f"topk n={n} k={k} desc={desc}")


def test_sort_large_vs_oracle_multikey():
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    n = 400_000
    t = pa.table({
        "a": pa.array(rng.integers(-3, 3, n).astype(np.int64), mask=rng.random(n) < 0.05),
        "f": pa.array(np.round(rng.normal(0, 5, n), 1), mask=rng.random(n) < 0.05),
        "u": pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2)),
        "rowid": pa.array(np.arange(n, dtype=np.int64)),
    })
    for cols, orders in [(["a", "f"], [0, 1]), (["f", "a"], [1, 0]), (["u"], [1])]:
        got = gpu_sort(t, cols, orders)
        s = O.OracleSort(cols, orders)
        for b in t.to_batches():
            s.next(b)
        util.assert_batches_equal(got, s.sorted(), what=f"{cols} {orders}")


def test_projection_matches_numpy_golden():
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    case = MAN["project"][0]
    table = util.read_ipc(case["input"]).combine_chunks()
    expected = util.read_ipc(case["expected"]).combine_chunks()
    dev = {n: DeviceColumn.from_arrow(table.column(n)) for n in table.schema.names}

    def conv(e):
        return tuple(conv(x) for x in e) if isinstance(e, list) else e

    for name, expr in case["exprs"].items():
        got = ops.project(conv(expr), dev, length=table.num_rows).to_arrow()
        util.assert_col_equal(got, expected.column(name), f"project {name}")  # bit-exact incl. float results


def test_projection_nulls_and_scalars():
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    rng = np.random.default_rng(9)
    n = 10_000
    i = pa.array(rng.integers(-100, 100, n).astype(np.int64), mask=rng.random(n) < 0.1)
    v = pa.array(rng.normal(size=n), mask=rng.random(n) < 0.1)
    dev = {"i": DeviceColumn.from_arrow(i), "v": DeviceColumn.from_arrow(v)}
    ni = i.to_numpy(zero_copy_only=False)   # NULL -> NaN, float64 (record_batch.py:112-118)
    nv = v.to_numpy(zero_copy_only=False)
    with np.errstate(all="ignore"):
        for expr, ref in [(("add", "i", 1), ni + 1), (("mul", "v", "i"), nv * ni), (("mod", "i", 7), np.mod(ni, 7)),
                          (("div", "v", 0), nv / 0)]:
            got = ops.project(expr, dev, length=n).to_numpy()
            assert np.array_equal(got.view(np.uint64), np.asarray(ref, dtype=np.float64).view(np.uint64)), expr
    got = ops.project(("add", ("mul", 2, 3), 1), {}, length=5).to_numpy()   # scalar-only: np.repeat (algebra.py:77-87)
    assert got.tolist() == [7] * 5


def test_project_many_equals_single_expression_kernels():
    """A whole SELECT list in one kernel (vnm_project_multi) == one kernel per expression, bit for bit; ragged
    lengths exercise the four-rows-per-lane tail; > 16 outputs exercise the host-side program packing."""
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    case = MAN["project"][0]
    table = util.read_ipc(case["input"]).combine_chunks()
    expected = util.read_ipc(case["expected"]).combine_chunks()

    def conv(e):
        return tuple(conv(x) for x in e) if isinstance(e, list) else e

    names = list(case["exprs"])
    exprs = [conv(case["exprs"][n]) for n in names]
    for length in (table.num_rows, 1, 255, 1025):
        t = table.slice(0, length)
        dev = {n: DeviceColumn.from_arrow(t.column(n)) for n in t.schema.names}
        outs = ops.project_many(exprs, dev, length=length)
        assert len(outs) == len(exprs)
        for name, col in zip(names, outs):
            util.assert_col_equal(col.to_arrow(), expected.column(name).slice(0, length), f"project_many {name} n={length}")
    # predicates and arithmetic in the same program, more outputs than one program holds
    rng = np.random.default_rng(4)
    n = 3000
    a = rng.integers(-50, 50, n).astype(np.int64)
    v = rng.normal(size=n)
    dev = {"a": DeviceColumn.from_numpy(a), "v": DeviceColumn.from_numpy(v)}
    exprs = [("add", "a", k) for k in range(20)] + [("gt", "v", 0.0), ("mul", "v", "a")]
    outs = ops.project_many(exprs, dev, length=n)
    for k in range(20):
        assert np.array_equal(outs[k].to_numpy(), a + k)
    assert np.array_equal(outs[20].to_numpy().astype(bool), v > 0.0)
    assert np.array_equal(outs[21].to_numpy().view(np.uint64), (v * a).view(np.uint64))


def _rand_col(rng, n, kind, null_p, card):
    """A column of `kind` with about `card` distinct values (ties matter for stability) and NULLs / NaNs."""
    base = rng.integers(0, max(card, 1), n)
    mask = (rng.random(n) < null_p) if null_p else None
    if kind == "f64":
        a = base.astype(np.float64) * 0.37 - card * 0.1
        a[rng.random(n) < 0.01] = np.nan
        a[rng.random(n) < 0.01] = -0.0
        a[rng.random(n) < 0.005] = np.inf
    elif kind == "f32":
        a = (base.astype(np.float32) * np.float32(0.5)) - np.float32(3)
    elif kind == "i64":
        a = (base.astype(np.int64) - card // 2) * 1_000_003
    elif kind == "i32":
        a = (base - card // 2).astype(np.int32)
    elif kind == "u16":
        a = (base % 65536).astype(np.uint16)
    else:
        a = (base.astype(np.uint64) * np.uint64(9_007_199_254_740_993)) % np.uint64(2**64 - 1)
    return pa.array(a, mask=mask)


import os


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VNM_FUZZ_SEEDS", "40")))))
def test_random_sorts_vs_oracle(seed):
    """Seeded differential test: 1-3 sort keys of random types (floats with NaN / -0.0 / inf, NULLs, narrow ints,
    uint64 beyond 2^63), random directions, many ties (stability), optional LIMIT, a payload column taken along."""
    from oracle import oracle as O
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([1, 2, 63, 1000, 70_000, 150_000, 260_000]))
    nk = int(rng.choice([1, 1, 2, 3]))
    cols, names, orders = {}, [], []
    for j in range(nk):
        kind = str(rng.choice(["f64", "f32", "i64", "i32", "u16", "u64"]))
        card = int(rng.choice([1, 3, 50, 5000, 10**6]))
        cols[f"s{j}"] = _rand_col(rng, n, kind, 0.03 if rng.random() < 0.4 else 0.0, card)
        names.append(f"s{j}")
        orders.append(int(rng.integers(0, 2)))
    cols["rowid"] = pa.array(np.arange(n, dtype=np.int64))
    t = pa.table(cols)
    limit = int(rng.choice([0, 0, 1, 10, 5000])) if n > 100 else 0
    limit = min(limit, n)
    got = gpu_sort(t, names, orders, limit=limit)
    s = O.OracleSort(names, orders)
    for b in t.to_batches():
        s.next(b)
    exp = s.sorted()
    if limit:
        exp = exp.slice(0, limit)
    util.assert_batches_equal(got, exp, what=f"seed {seed}: n={n} keys {[str(t.schema.field(c).type) for c in names]} "
                                             f"orders {orders} limit {limit}")


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VNM_FUZZ_SEEDS", "40")))))
def test_random_expressions_vs_numpy(seed):
    """Seeded differential test of the fused projection: random arithmetic trees over int64 / float64 columns (one
    of them with NULLs, which reach NumPy as float64 NaN, record_batch.py:112-118) and int / float literals,
    evaluated the way the reference does it -- one NumPy ufunc per node (expressions.py:13-24)."""
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([1, 255, 1024, 1025, 5000, 70_001]))
    a = rng.integers(-1000, 1000, n).astype(np.int64)
    b = rng.integers(0, 2**12, n).astype(np.float64) / 16.0 - 50.0
    cvals = rng.integers(-50, 50, n).astype(np.int64)
    cmask = rng.random(n) < 0.1
    arrow = {"a": pa.array(a), "b": pa.array(b), "c": pa.array(cvals, mask=cmask)}
    dev = {k: DeviceColumn.from_arrow(v) for k, v in arrow.items()}
    # only a column that really has NULLs is converted (null_count > 0, record_batch.py:112-118)
    npv = {"a": a, "b": b, "c": np.where(cmask, np.nan, cvals.astype(np.float64)) if cmask.any() else cvals}
    ufunc = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "mod": np.mod}

    def gen(depth):
        r = rng.random()
        if depth == 0 or r < 0.25:
            return str(rng.choice(["a", "b", "c"])) if rng.random() < 0.75 else (
                int(rng.integers(-5, 6)) if rng.random() < 0.5 else float(rng.choice([0.5, -2.25, 3.0])))
        if r < 0.35:
            return ("neg", gen(depth - 1))
        return (str(rng.choice(list(ufunc))), gen(depth - 1), gen(depth - 1))

    def has_col(e):
        return isinstance(e, str) or (isinstance(e, tuple) and any(has_col(x) for x in e[1:]))

    def ev(e):
        if isinstance(e, str):
            return npv[e]
        if isinstance(e, (int, float)):
            return e
        if e[0] == "neg":
            return np.negative(ev(e[1]))
        return ufunc[e[0]](ev(e[1]), ev(e[2]))

    exprs = []
    while len(exprs) < 4:
        e = gen(3)
        if isinstance(e, tuple) and has_col(e):
            exprs.append(e)
    with np.errstate(all="ignore"):
        refs = [np.asarray(ev(e)) for e in exprs]
    used = {k: dev[k] for k in ("a", "b", "c")}
    outs = ops.project_many(exprs, used, length=n)
    for e, ref, out in zip(exprs, refs, outs):
        got = out.to_numpy()
        ref = np.broadcast_to(ref, (n,))
        assert got.dtype == ref.dtype, (seed, e, got.dtype, ref.dtype)
        if ref.dtype.kind == "f":
            both_nan = np.isnan(got) & np.isnan(ref)
            same = (got.view(np.uint64) == np.ascontiguousarray(ref).view(np.uint64)) | both_nan
        else:
            same = got == ref
        assert same.all(), f"seed {seed}: {e}: row {int(np.argmin(same))}: {got[np.argmin(same)]!r} vs {ref[np.argmin(same)]!r}"


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VNM_FUZZ_SEEDS", "40")))))
def test_random_expressions_narrow_types_vs_numpy(seed):
    """The same differential test over EVERY numeric Arrow width (int8..uint64, float32, float64; some columns with
    NULLs): result types, wraparound of narrow integers, float32 arithmetic, weak Python literals, comparisons
    (int64 vs uint64 exactly, float32 against a literal in float32), IN lists (np.isin: strong int64 / float64) must be
    what NumPy (vinum/core/expressions.py:13-48 dispatches to it node by node) produces, bit for bit."""
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    rng = np.random.default_rng(91000 + seed)
    n = int(rng.choice([1, 257, 4096, 4097, 30_001]))
    makers = {
        "i8": lambda: rng.integers(-128, 128, n).astype(np.int8), "i16": lambda: rng.integers(-2**15, 2**15, n).astype(np.int16),
        "i32": lambda: rng.integers(-2**31, 2**31, n).astype(np.int32), "i64": lambda: rng.integers(-2**40, 2**40, n).astype(np.int64),
        "u8": lambda: rng.integers(0, 256, n).astype(np.uint8), "u16": lambda: rng.integers(0, 2**16, n).astype(np.uint16),
        "u32": lambda: rng.integers(0, 2**32, n).astype(np.uint32), "u64": lambda: rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2) + np.uint64(1),
        "f32": lambda: (rng.normal(0, 50, n)).astype(np.float32), "f64": lambda: rng.normal(0, 50, n),
    }
    names = list(rng.choice(list(makers), size=4, replace=False))
    arrow, npv = {}, {}
    for nm in names:
        vals = makers[nm]()
        mask = (rng.random(n) < 0.1) if rng.random() < 0.25 else None
        arr = pa.array(vals, mask=mask)
        arrow[nm] = arr
        # what the reference hands to NumPy: to_numpy, NULL -> NaN (ints become float64, float32 stays float32)
        npv[nm] = arr.to_numpy(zero_copy_only=False) if arr.null_count else vals
    dev = {k: DeviceColumn.from_arrow(v) for k, v in arrow.items()}
    arith = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "mod": np.mod}
    bits = {"band": np.bitwise_and, "bor": np.bitwise_or, "bxor": np.bitwise_xor}
    cmps = {"eq": lambda x, y: x == y, "ne": lambda x, y: x != y, "gt": lambda x, y: x > y, "ge": lambda x, y: x >= y,
            "lt": lambda x, y: x < y, "le": lambda x, y: x <= y}

    def lit():
        return int(rng.integers(0, 100)) if rng.random() < 0.6 else float(rng.choice([0.5, -2.25, 0.1, 3.0]))

    def gen(depth):
        r = rng.random()
        if depth == 0 or r < 0.3:
            return str(rng.choice(names)) if rng.random() < 0.7 else lit()
        if r < 0.38:
            return (str(rng.choice(["neg", "bnot"])), gen(depth - 1))
        pool = list(arith) * 3 + list(bits)
        return (str(rng.choice(pool)), gen(depth - 1), gen(depth - 1))

    def has_col(e):
        return isinstance(e, str) or (isinstance(e, tuple) and any(has_col(x) for x in e[1:]))

    def ev(e):
        if isinstance(e, str):
            return npv[e]
        if isinstance(e, (int, float)):
            return e
        if e[0] == "neg":
            return np.negative(ev(e[1]))
        if e[0] == "bnot":
            return ~ev(e[1])
        if e[0] in ("in", "not_in"):
            return np.isin(ev(e[1]), list(e[2]), invert=e[0] == "not_in")
        f = arith.get(e[0]) or bits.get(e[0]) or cmps[e[0]]
        return f(ev(e[1]), ev(e[2]))

    cases = []
    tries = 0
    while len(cases) < 6 and tries < 400:
        tries += 1
        e = gen(3)
        if not (isinstance(e, tuple) and has_col(e)):
            continue
        kind = rng.random()
        if kind < 0.3:
            e = (str(rng.choice(list(cmps))), e, gen(1))
        elif kind < 0.4:
            e = (str(rng.choice(["in", "not_in"])), str(rng.choice(names)), tuple(lit() for _ in range(3)))
        try:
            with np.errstate(all="ignore"), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = np.asarray(ev(e))
        except (TypeError, OverflowError):
            # NumPy refuses (bitwise on floats, uint64 with a signed type, a literal out of range): so must we
            with pytest.raises(Exception):
                ops.project_many([e], dev, length=n)
            continue
        cases.append((e, np.broadcast_to(ref, (n,))))
    outs = ops.project_many([e for e, _ in cases], dev, length=n)
    for (e, ref), out in zip(cases, outs):
        got = out.to_numpy()
        if ref.dtype == np.bool_:
            assert out.arrow_type == pa.uint8() and (got.astype(bool) == ref).all(), (seed, e)
            continue
        assert got.dtype == ref.dtype, (seed, e, got.dtype, ref.dtype)
        if ref.dtype.kind == "f":
            it = np.uint64 if ref.dtype == np.float64 else np.uint32
            same = (got.view(it) == np.ascontiguousarray(ref).view(it)) | (np.isnan(got) & np.isnan(ref))
        else:
            same = got == ref
        assert same.all(), f"seed {seed}: {e}: row {int(np.argmin(same))}: {got[np.argmin(same)]!r} vs {ref[np.argmin(same)]!r}"


def test_out_of_range_literal_true_division_vs_numpy():
    """NumPy range-checks a Python integer against the column's type for + - * % (OverflowError) but not for true division,
    which runs in float64 for every integer width (fuzz seed 73 of the 400-seed campaign: `~59 / uint64`)."""
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    cols = {"u64": np.array([5, 7, 2**63 + 3], np.uint64), "i8": np.array([5, -7, 100], np.int8), "u8": np.array([1, 2, 255], np.uint8)}
    dev = {k: DeviceColumn.from_arrow(pa.array(v)) for k, v in cols.items()}
    good = [(("div", "u64", -60), np.divide(cols["u64"], -60)), (("div", -60, "u64"), np.divide(-60, cols["u64"])),
            (("div", "i8", 300), np.divide(cols["i8"], 300)), (("div", ("bnot", 59), "u8"), np.divide(~59, cols["u8"])),
            (("div", 300, "i8"), np.divide(300, cols["i8"]))]
    outs = ops.project_many([e for e, _ in good], dev, length=3)
    for (e, ref), out in zip(good, outs):
        got = out.to_numpy()
        assert got.dtype == np.float64 and (got.view(np.uint64) == ref.view(np.uint64)).all(), (e, got, ref)
    for e in [("add", "u64", -60), ("mod", "i8", 300), ("mul", -1, "u8"), ("sub", 300, "i8")]:
        with pytest.raises(Exception, match="out of bounds"):
            ops.project_many([e], dev, length=3)


@pytest.mark.parametrize("kind", ["f64", "i64_desc", "u64", "negzero", "nan", "null", "i32", "two_keys"])
def test_full_sort_returns_its_first_key_sorted(kind):
    """vnm_sort_indices_keyed: for an 8-byte first key without NULL / NaN / -0.0 the sorted key column comes out of the last radix
    pass (rebuilt from the codes) and must equal the gather bit for bit; with a NaN, a NULL, a -0.0 (whose code is that of +0.0)
    or a narrower key the sort declines and the caller gathers."""
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    rng = np.random.default_rng(len(kind))
    n = 300_001
    order = 1 if kind == "i64_desc" else 0
    mask = None
    if kind in ("f64", "negzero", "nan", "null", "two_keys"):
        v = np.round(rng.normal(0, 50, n), 2)
        v[v == 0] = 0.0
        if kind == "negzero":
            v[17] = -0.0
        if kind == "nan":
            v[23] = np.nan
        if kind == "null":
            mask = np.zeros(n, bool); mask[5] = True
    elif kind == "u64":
        v = rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    elif kind == "i32":
        v = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    else:
        v = rng.integers(-2**62, 2**62, n).astype(np.int64)
    arr = pa.array(v, mask=mask)
    keys = [DeviceColumn.from_arrow(arr)]
    orders = [order]
    if kind == "two_keys":
        keys.append(DeviceColumn.from_arrow(pa.array(rng.integers(0, 5, n).astype(np.int64))))
        orders.append(1)
    idx, sk = ops.sort_indices_keyed(keys, orders)
    gathered = ops.take(keys[0], idx, n).to_arrow()
    if kind in ("negzero", "nan", "null", "i32"):
        assert sk is None
    else:
        assert sk is not None
        util.assert_col_equal(sk.to_arrow(), gathered, kind)
        got = sk.to_numpy()
        assert (np.diff(got.astype(np.float64)) <= 0).all() if order else (np.diff(got.astype(np.float64)) >= 0).all()


@pytest.mark.parametrize("shape", ["distinct_first_key", "ties_on_first_key", "few_values_first_key", "specials"])
@pytest.mark.parametrize("k", [10, 3000])
def test_topk_multi_key_equals_full_sort_prefix(shape, k):
    """ORDER BY a [DESC], b, c LIMIT K: candidates are selected on the first key alone (everything that beats the threshold, ties
    included) and only they are sorted by all keys; the first K rows must be those of the stable full sort (the oracle), also when
    the first key ties heavily (the candidate set outgrows its buffer and the full sort takes over) or holds NaN / NULL."""
    from oracle import oracle as O
    rng = np.random.default_rng(len(shape) * 7 + k)
    n = 1_200_000
    if shape == "distinct_first_key":
        a = rng.permutation(n).astype(np.float64)
    elif shape == "ties_on_first_key":
        a = np.round(rng.normal(0, 1000, n), 0)            # ~6000 distinct values: ties of ~200 rows each
    elif shape == "few_values_first_key":
        a = rng.integers(0, 4, n).astype(np.float64)       # every row ties: the fast path must hand over
    else:
        a = np.round(rng.normal(0, 1000, n), 0)
        a[rng.random(n) < 0.002] = np.nan
    mask = (rng.random(n) < 0.002) if shape == "specials" else None
    b = rng.integers(-50, 50, n).astype(np.int64)
    c = rng.integers(0, 2**31, n).astype(np.int32)
    t = pa.table({"rowid": pa.array(np.arange(n, dtype=np.int64)), "a": pa.array(a, mask=mask),
                  "b": pa.array(b, mask=(rng.random(n) < 0.01) if shape == "specials" else None), "c": pa.array(c)})
    for orders in ([1, 0, 1], [0, 1, 0]):
        got = gpu_sort(t, ["a", "b", "c"], orders, limit=k)
        s = O.OracleSort(["a", "b", "c"], orders)
        for bt in t.to_batches():
            s.next(bt)
        exp = s.sorted().slice(0, k)
        util.assert_batches_equal(got, exp, what=f"multi-key topk {shape} k={k} orders={orders}")


@pytest.mark.parametrize("case", ["f64_normal", "f64_desc_nan_negzero", "i64_few_dups", "u64_desc", "runs_of_100", "heavy_value", "heavy_values_and_nans_desc",
                                  "low_cardinality_declines", "tiny_buckets", "odd_size", "fanout_16", "fanout_64_forced", "fanout_512_forced",
                                  "nulls_asc", "nulls_desc_nan_heavy", "half_null_unaligned", "mostly_null_declines"])
def test_sample_sort_equals_the_lsd_sort(case, monkeypatch):
    """The sample sort of one 8-byte key (vnm_sort_sample.inc: splitters from a sorted sample, two ring scatters into 2^18 buckets,
    per-bucket LSD sort in LDS, rows of equal key by row id) must give the SAME row ids as the eight-pass LSD sort -- the order is
    total (key, then row id: Sort::Sorted is stable, sort.cpp:22-40) -- and the same rebuilt key column.  Rows with equal keys in
    short and long runs, NaN / -0.0, descending order, heavy values (side list), and data it declines (1000 distinct values)."""
    import torch
    from vinum_amd import _lib as L
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    rng = np.random.default_rng(77)
    n = 3_000_000 if case != "tiny_buckets" else 300_000
    order = L.ASC
    if case == "f64_normal":
        v = rng.normal(11.0, 3.0, n)
    elif case == "f64_desc_nan_negzero":
        v = rng.normal(0.0, 1.0, n); v[::977] = np.nan; v[5::1201] = -0.0; v[7::1201] = 0.0; order = L.DESC
    elif case == "i64_few_dups":
        v = rng.integers(0, n // 4, n).astype(np.int64) - n // 8
    elif case == "u64_desc":
        v = rng.integers(0, 2**63, n).astype(np.uint64) * 2 + rng.integers(0, 2, n).astype(np.uint64); order = L.DESC
    elif case == "runs_of_100":
        v = (rng.integers(0, n // 100, n) * 1_000_003).astype(np.int64)
    elif case == "heavy_value":                       # one value holds 10 % of the rows: its rows bypass the buckets (side list)
        v = rng.normal(0.0, 1.0, n); v[::10] = 0.25
    elif case == "heavy_values_and_nans_desc":        # several heavy values, the smallest and the largest key among them, and NaNs
        v = rng.normal(0.0, 1.0, n); v[::7] = 12.5; v[3::11] = -40.0; v[5::13] = np.nan; v[1::17] = 0.0; v[2::170] = -0.0; order = L.DESC
    elif case == "low_cardinality_declines":          # 1000 distinct values: more heavy codes than the side list takes -> LSD sort
        v = rng.integers(0, 1000, n).astype(np.float64)
    elif case == "odd_size":
        n = 2_999_999 + 4096 * 3 + 17; v = rng.normal(0.0, 1.0, n)
    elif case == "fanout_16":                          # the second level's fan-out follows n: 3.4e7 rows -> 512 x 16 buckets
        n = (1 << 25) + 12345; v = rng.normal(0.0, 1.0, n); v[::1_000_003] = np.nan; order = L.DESC
    elif case in ("fanout_64_forced", "fanout_512_forced"):
        monkeypatch.setenv("VNM_SSORT_L2", case.split("_")[1])
        v = rng.integers(-2**62, 2**62, n).astype(np.int64); v[::3] = v[1::3][:len(v[::3])]      # duplicates across rows
    else:
        v = rng.normal(0.0, 1.0, n)
    mask = None
    if case == "nulls_asc":                            # NULL rows: a class of the side list, behind every value, in row order
        mask = rng.random(n) < 0.001
    elif case == "nulls_desc_nan_heavy":
        v[::5] = 3.5; v[3::1009] = np.nan; v[7::2003] = -0.0; mask = rng.random(n) < 0.02; order = L.DESC
    elif case == "half_null_unaligned":
        v = rng.integers(-2**40, 2**40, n).astype(np.int64); mask = rng.random(n) < 0.5
    elif case == "mostly_null_declines":
        mask = rng.random(n) < 0.9995
    if mask is not None:
        arr = pa.array(v, mask=mask)
        if case == "half_null_unaligned":
            arr = arr.slice(3, n - 5); n = len(arr)    # an Arrow offset that is not a multiple of 8 (odd: no 16-byte pairs)
        col = DeviceColumn.from_arrow(arr)
    else:
        t = torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).cuda()
        col = DeviceColumn.from_torch(t)
        if v.dtype == np.uint64:
            col = DeviceColumn(col._values, None, 0, n, pa.uint64(), keep=t)
    monkeypatch.setenv("VNM_SORT_NO_SAMPLE", "1")
    ref_idx, ref_key = ops.sort_indices_keyed([col], [order])
    ref = torch.as_tensor(_RawI64(ref_idx.ptr, n), device="cuda").clone()
    monkeypatch.delenv("VNM_SORT_NO_SAMPLE")
    monkeypatch.setenv("VNM_SSORT_MIN_ROWS", "1000")
    import ctypes
    L.lib().vnm_set_profiling(1)
    got_idx, got_key = ops.sort_indices_keyed([col], [order])
    ms, local = ctypes.c_double(0), ctypes.c_int64(0)
    L.lib().vnm_profile_query(b"sort_local", ctypes.byref(ms), ctypes.byref(local))
    L.lib().vnm_set_profiling(0)
    assert (local.value >= 1) == (case not in ("low_cardinality_declines", "mostly_null_declines")), (case, local.value)   # the path under test ran
    got = torch.as_tensor(_RawI64(got_idx.ptr, n), device="cuda")
    assert bool(torch.equal(got, ref)), f"{case}: {int((got != ref).sum())} positions differ"
    assert (got_key is None) == (ref_key is None)
    if got_key is not None:
        a = torch.as_tensor(_RawI64(got_key.values_ptr, n), device="cuda")
        b = torch.as_tensor(_RawI64(ref_key.values_ptr, n), device="cuda")
        assert bool(torch.equal(a, b))


class _RawI64:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("desc", [False, True])
def test_distributed_sample_sort_simulated(world, desc):
    """distributed.sample_sort_exchange on one GPU, against the ORACLE: `world` shards play the ranks (contiguous row ranges); the
    splitters come from the union of their samples; every rank partitions its rows by owner ON THE DEVICE (vnm_partition_by_owner --
    checked against a stable argsort of the owners); the blocks are routed exactly as the all_to_all would deliver them (by owner, in
    source-rank order); every owner sorts what it got with the library's stable single-key sort over the order codes.  The ranks'
    slices, concatenated in rank order, must be Arrow's sort_indices of the whole column (Sort::Sorted, sort.cpp:22-44: stable, NaN
    after every number in both directions) -- and the single-GPU sort agrees with that, too."""
    import pyarrow.compute as pc
    import torch
    from vinum_amd import _lib as L
    from vinum_amd import distributed as D
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    rng = np.random.default_rng(5)
    n = 1_200_000
    v = rng.normal(0.0, 1.0, n); v[::997] = np.nan; v[5::1201] = -0.0; v[7::1201] = 0.0; v[::11] = np.round(v[::11], 2)
    oracle = pc.sort_indices(pa.table({"x": pa.array(v)}), sort_keys=[("x", "descending" if desc else "ascending")])   # (NaN last in both directions; -0.0 == +0.0: stable)
    ref = torch.from_numpy(oracle.to_numpy().astype(np.int64)).cuda()
    tv = torch.from_numpy(v).cuda()
    single = ops.sort_indices([DeviceColumn.from_torch(tv)], [L.DESC if desc else L.ASC])
    assert bool(torch.equal(torch.as_tensor(_RawI64(single.ptr, n), device="cuda"), ref)), "single-GPU sort != Arrow sort_indices"
    bounds = np.linspace(0, n, world + 1).astype(int)
    codes = [D._order_key(tv[bounds[r]:bounds[r + 1]], desc) for r in range(world)]
    splitters = D.ssort_splitters([D.ssort_sample(c, 4096) for c in codes], world)
    parts = []
    for r in range(world):
        order, counts = D.partition_by_owner(codes[r], splitters, world)
        own = D.ssort_owner(codes[r], splitters)
        assert bool(torch.equal(order, torch.argsort(own, stable=True))), f"rank {r}: the device partition is not the stable partition by owner"
        assert bool(torch.equal(counts, torch.bincount(own, minlength=world)))
        parts.append((order, torch.cumsum(counts, 0) - counts, counts))
    got = []
    for o in range(world):
        blocks_c, blocks_i = [], []
        for r in range(world):                                  # what owner o receives: source-rank order, row order inside
            order, starts, counts = parts[r]
            sel = order[int(starts[o]):int(starts[o]) + int(counts[o])]
            blocks_c.append(codes[r][sel])
            blocks_i.append(sel + int(bounds[r]))
        rc, rid = torch.cat(blocks_c).contiguous(), torch.cat(blocks_i).contiguous()
        if len(rc):
            perm_buf = ops.sort_indices([DeviceColumn.from_torch(rc)], [L.ASC])      # stable: ties keep the (rank, row) = id order
            perm = torch.as_tensor(_RawI64(perm_buf.ptr, len(rc)), device="cuda")
            got.append(rid[perm])
    got = torch.cat(got)
    assert bool(torch.equal(got, ref)), f"world {world} desc {desc}: {int((got != ref).sum())} positions differ from Arrow's sort_indices"


@pytest.mark.parametrize("case", ["f32_normal", "f32_desc_nan_negzero", "i32_few_dups", "u32_desc", "i32_heavy_value", "f32_offset"])
def test_sample_sort_of_a_four_byte_key_equals_the_lsd_sort(case, monkeypatch):
    """Round 5: one float32 / int32 / uint32 sort key without NULLs is widened to 8 bytes (float32 -> float64 is exact and order-preserving,
    NaNs and signed zeros included) and ordered by the sample sort; the row ids must be those of the LSD passes over the original
    4-byte column (the order is total: key, then row id -- Sort::Sorted is stable, sort.cpp:22-40)."""
    import ctypes
    import torch
    from vinum_amd import _lib as L
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    rng = np.random.default_rng(len(case) * 7)
    n = 2_000_003
    order = L.ASC
    if case == "f32_normal":
        v = rng.normal(11.0, 3.0, n).astype(np.float32)
    elif case == "f32_desc_nan_negzero":
        v = rng.normal(0.0, 1.0, n).astype(np.float32); v[::977] = np.nan; v[5::1201] = -0.0; v[7::1201] = 0.0; order = L.DESC
    elif case == "i32_few_dups":
        v = (rng.integers(0, n // 4, n) - n // 8).astype(np.int32)
    elif case == "u32_desc":
        v = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32); order = L.DESC
    elif case == "i32_heavy_value":
        v = rng.integers(-2**31, 2**31, n).astype(np.int32); v[::9] = 12345
    else:
        v = rng.normal(0.0, 5.0, n + 7).astype(np.float32)
    arr = pa.array(v)
    if case == "f32_offset":
        arr = arr.slice(5, n)
    col = DeviceColumn.from_arrow(arr)
    monkeypatch.setenv("VNM_SSORT_MIN_ROWS", "1000")
    monkeypatch.setenv("VNM_SORT_NO_WIDEN", "1")
    ref_idx = ops.sort_indices([col], [order])
    ref = torch.as_tensor(_RawI64(ref_idx.ptr, n), device="cuda").clone()
    monkeypatch.delenv("VNM_SORT_NO_WIDEN")
    L.lib().vnm_set_profiling(1)
    got_idx = ops.sort_indices([col], [order])
    ms, local = ctypes.c_double(0), ctypes.c_int64(0)
    L.lib().vnm_profile_query(b"sort_local", ctypes.byref(ms), ctypes.byref(local))
    L.lib().vnm_set_profiling(0)
    assert local.value >= 1, case                        # the sample sort ran
    got = torch.as_tensor(_RawI64(got_idx.ptr, n), device="cuda")
    assert bool(torch.equal(got, ref)), f"{case}: {int((got != ref).sum())} positions differ"
    # ... and the order is right (against NumPy's stable sort of the same keys; NaN last in both directions)
    keys = arr.to_numpy()
    if order == L.ASC and not np.isnan(keys.astype(np.float64)).any():
        assert (got.cpu().numpy() == np.argsort(keys, kind="stable")).all()
