"""GPU parity: fused compare+compact filter (HIP, through the C ABI) vs the oracle / golden vectors.
Bit-exact (integer / byte / index work)."""
import numpy as np
import pyarrow as pa
import pytest

from tests import util

pytestmark = pytest.mark.gpu

OPS = {"eq": "==", "ne": "!=", "gt": ">", "ge": ">=", "lt": "<", "le": "<="}


def _gpu_filter(batch: pa.RecordBatch, column, op, literal) -> pa.RecordBatch:
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    cols = [DeviceColumn.from_arrow(c) for c in batch.columns]
    pred = cols[batch.schema.get_field_index(column)]
    outs, k = ops.filter_cmp(pred, op, literal, cols)
    return pa.RecordBatch.from_arrays([o.to_arrow() for o in outs], names=batch.schema.names)


@pytest.mark.parametrize("case", util.manifest()["filter"], ids=lambda c: c["name"])
def test_filter_matches_reference_golden(case):
    table = util.read_ipc(case["input"]).combine_chunks()
    exp_batches = util.read_ipc_batches(case["expected"])
    lit = float(case["literal"]) if case["literal_is_float"] else int(case["literal"])
    for (off, ln), exp in zip(case["slices"], exp_batches):
        batch = table.slice(off, ln).to_batches()[0]
        got = _gpu_filter(batch, case["column"], OPS[case["op"]], lit)
        util.assert_batches_equal(got, exp, what=case["name"])


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 4095, 4096, 4097, 100000, 1 << 20])
@pytest.mark.parametrize("nulls,offset", [(False, 0), (True, 0), (True, 3), (False, 5)])
def test_filter_edges_vs_oracle(n, nulls, offset):
    from oracle import oracle as O
    rng = np.random.default_rng(n + offset)
    total = n + offset + 7
    x = rng.integers(0, 2**14, total).astype(np.float64) / 128.0
    y = rng.integers(-10**9, 10**9, total).astype(np.int64)
    z = rng.integers(0, 255, total).astype(np.uint8)
    mx = (rng.random(total) < 0.1) if nulls else None
    my = (rng.random(total) < 0.2) if nulls else None
    batch = pa.RecordBatch.from_arrays(
        [pa.array(x, mask=mx), pa.array(y, mask=my), pa.array(z)], names=["x", "y", "z"]).slice(offset, n)
    for col, op, lit in [("x", ">", 64.0), ("x", "<=", 1.0), ("y", ">", 0), ("y", "!=", 5), ("z", "<", 100)]:
        got = _gpu_filter(batch, col, op, lit)
        ocode = {">": O.GT, "<=": O.LE, "!=": O.NE, "<": O.LT}[op]
        mask = O.cmp_mask(batch.column(batch.schema.get_field_index(col)), ocode, lit)
        exp = O.filter_batch(batch, mask)
        util.assert_batches_equal(got, exp, what=f"n={n} {col}{op}{lit} nulls={nulls} off={offset}")


def test_filter_selectivity_sweep_large():
    """Size-independent properties at a large size: count == popcount(mask), order preserved, values exact."""
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    rng = np.random.default_rng(7)
    n = 20_000_003
    x = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    col = DeviceColumn.from_numpy(x)
    for thr in [-1.0, 1.28, 64.0, 126.7, 1e9]:
        outs, k = ops.filter_cmp(col, ">", thr, [col])
        ref = x[x > thr]
        assert k == len(ref)
        got = outs[0].to_numpy()
        assert np.array_equal(got, ref)


def test_filter_emit_null_mask():
    from oracle import oracle as O
    from vinum_amd.device import DeviceColumn, DeviceBuffer
    from vinum_amd import ops
    rng = np.random.default_rng(5)
    n = 70001
    batch = pa.RecordBatch.from_arrays(
        [pa.array(rng.normal(size=n), mask=rng.random(n) < 0.1), pa.array(rng.integers(0, 9, n))], names=["x", "y"])
    m = (rng.random(n) < 0.5)
    mv = (rng.random(n) < 0.9)
    cols = [DeviceColumn.from_arrow(c) for c in batch.columns]
    outs, k = ops.filter_mask(DeviceBuffer.from_host(m.astype(np.uint8)), DeviceBuffer.from_host(mv.astype(np.uint8)), n, cols)
    got = pa.RecordBatch.from_arrays([o.to_arrow() for o in outs], names=batch.schema.names)
    util.assert_batches_equal(got, O.filter_batch(batch, m, mv), what="emit_null")


import os


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VNM_FUZZ_SEEDS", "40")))))
def test_random_filters_vs_oracle(seed):
    """Seeded differential test: predicate column / payload columns of random types, NULLs, Arrow slice offsets,
    every comparison operator, int and float literals, sizes around the tile boundaries."""
    from oracle import oracle as O
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([0, 1, 65, 8191, 8192, 8193, 40_000, 131_072, 300_001]))
    offset = int(rng.choice([0, 0, 1, 3, 8, 13]))
    total = n + offset + 5
    makers = {
        "f64": lambda: rng.integers(0, 2**14, total).astype(np.float64) / 128.0,
        "f32": lambda: (rng.integers(0, 2**10, total) / 8.0).astype(np.float32),
        "i64": lambda: rng.integers(-10**9, 10**9, total).astype(np.int64),
        "i32": lambda: rng.integers(-1000, 1000, total).astype(np.int32),
        "u8": lambda: rng.integers(0, 255, total).astype(np.uint8),
        "u64": lambda: rng.integers(0, 2**63, total).astype(np.uint64) * np.uint64(2),
    }
    ncols = int(rng.integers(1, 5))
    arrays, names = [], []
    for j in range(ncols):
        kind = str(rng.choice(list(makers)))
        a = makers[kind]()
        mask = (rng.random(total) < 0.15) if rng.random() < 0.4 else None
        arrays.append(pa.array(a, mask=mask))
        names.append(f"c{j}")
    batch = pa.RecordBatch.from_arrays(arrays, names=names).slice(offset, n)
    col = names[int(rng.integers(0, ncols))]
    arr = batch.column(batch.schema.get_field_index(col))
    op = str(rng.choice(["==", "!=", ">", ">=", "<", "<="]))
    if pa.types.is_floating(arr.type):
        lit = float(rng.choice([1.0, 64.0, 100.5]))
    else:
        lit = int(rng.choice([0, 5, 100])) if rng.random() < 0.8 else float(rng.choice([0.5, 99.5]))
    got = _gpu_filter(batch, col, op, lit)
    ocode = {"==": O.EQ, "!=": O.NE, ">": O.GT, ">=": O.GE, "<": O.LT, "<=": O.LE}[op]
    exp = O.filter_batch(batch, O.cmp_mask(arr, ocode, lit))
    util.assert_batches_equal(got, exp, what=f"seed {seed}: n={n} off={offset} {col}:{arr.type} {op} {lit} "
                                             f"cols {[str(a.type) for a in arrays]}")


@pytest.mark.parametrize("scheme", ["flat_gives_up", "cooperative_only", "flat"])
def test_filter_launch_schemes(scheme, monkeypatch):
    """The filter is launched one workgroup per tile with BOUNDED look-backs first and repeated with cooperative persistent
    workgroups if one of them gave up.  A spin limit of zero polls makes (nearly) every tile give up: the first attempt's output
    is discarded and the cooperative repeat must deliver the exact result."""
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    if scheme == "flat_gives_up":
        monkeypatch.setenv("VNM_FILTER_SPIN_LIMIT", "1")
        monkeypatch.setenv("VNM_FILTER_SLEEP", "0")
    elif scheme == "cooperative_only":
        monkeypatch.setenv("VNM_FILTER_PERSIST", "1")
    rng = np.random.default_rng(11)
    n = 30_000_001
    x = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    y = rng.integers(-2**40, 2**40, n).astype(np.int64)
    cx, cy = DeviceColumn.from_numpy(x), DeviceColumn.from_numpy(y)
    import ctypes
    from vinum_amd import _lib as L
    L.lib().vnm_set_profiling(1)
    for thr in [1.28, 64.0, 126.7]:
        outs, k = ops.filter_cmp(cx, ">", thr, [cx, cy])
        keep = x > thr
        assert k == int(keep.sum())
        assert np.array_equal(outs[0].to_numpy(), x[keep]) and np.array_equal(outs[1].to_numpy(), y[keep])
    ms, retries = ctypes.c_double(0), ctypes.c_int64(0)
    L.lib().vnm_profile_query(b"filter_retry", ctypes.byref(ms), ctypes.byref(retries))
    L.lib().vnm_set_profiling(0)
    # the retry counter: every filter repeated when the look-backs give up at once, none otherwise
    assert retries.value == (3 if scheme == "flat_gives_up" else 0), retries.value


@pytest.mark.parametrize("n", [0, 1, 4095, 8192, 8193, 100_003, 1 << 21])
@pytest.mark.parametrize("kind", ["int64", "timestamp", "int64_with_payload", "int64_nullable"])
@pytest.mark.parametrize("op", [">", "<=", "==", "!="])
def test_int64_predicate_hot_path(n, kind, op):
    """Round 5: an int64 (or timestamp / date64: int64 storage) predicate column without NULLs against an INTEGER literal takes the same
    register-resident 16-byte-load kernel as float64 (before: the generic per-row path, 2.5 x the time).  Survivors and payload columns
    equal NumPy's boolean indexing; negative values, the literal at the extremes, a nullable column (generic path) for comparison."""
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    rng = np.random.default_rng(n + len(kind) + len(op))
    x = rng.integers(-50, 50, n).astype(np.int64) * 1_000_003
    lit = int(x[n // 2]) if n else 7
    mask = (rng.random(n) < 0.1) if kind == "int64_nullable" else None
    t = pa.timestamp("us") if kind == "timestamp" else pa.int64()
    pred_arr = pa.array(x, mask=mask).cast(t)
    pay = pa.array(rng.integers(0, 1000, n).astype(np.int32))
    cols = [DeviceColumn.from_arrow(pred_arr)] + ([DeviceColumn.from_arrow(pay)] if kind == "int64_with_payload" else [])
    outs, k = ops.filter_cmp(cols[0], op, lit, cols)
    keep = {">": x > lit, "<=": x <= lit, "==": x == lit, "!=": x != lit}[op]
    if mask is not None:          # a NULL compares like NaN (the reference hands NumPy NaNs there): only `!=` is true for it
        keep = (keep | mask) if op == "!=" else (keep & ~mask)
    assert k == int(keep.sum())
    got = outs[0].to_arrow().cast(pa.int64())
    valid = np.ones(k, bool) if mask is None else ~mask[keep]
    assert got.null_count == int((~valid).sum())
    assert (got.fill_null(0).to_numpy(zero_copy_only=False)[valid] == x[keep][valid]).all()
    if kind == "int64_with_payload":
        assert (outs[1].to_arrow().to_numpy() == pay.to_numpy()[keep]).all()
