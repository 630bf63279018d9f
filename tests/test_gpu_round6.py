"""Round 6 parity: the data-dependent fast routes of the dense path -- fixed-point entry words (DPartArgs::fx_q) and the final pass
without compensation words (exact_track) -- against the oracle, with every way out of them: a value the sample did not announce
(finer fraction, larger magnitude, -0.0, NaN / Inf), totals beyond 2^53 quanta, batches that join or cannot join a pending pass."""
import ctypes
import math
import os

import numpy as np
import pyarrow as pa
import pytest

from tests import util
from tests.test_gpu_agg import gpu_aggregate

pytestmark = pytest.mark.gpu


def _routes():
    from vinum_amd import _lib as L
    lib = L.lib()
    need = lib.vnm_route_counts(None, 0)
    buf = ctypes.create_string_buffer(int(need) + 16)
    lib.vnm_route_counts(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        k, _, v = line.rpartition("=")
        out[k] = int(v)
    return out


def _took(before, name):
    return _routes().get(name, 0) - before.get(name, 0)


def _oracle(funcs, batches, pred=None):
    from oracle import oracle as O
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in batches:
        if pred is not None:
            b = O.filter_batch(b, O.cmp_mask(b.column(1), O.GT, pred[2]))
        o.next(b)
    return o.result()


def _hot_funcs():
    from oracle import oracle as O
    return [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT_STAR, "", "n")]


@pytest.mark.parametrize("groups", [1_500_000, 12_000_000])          # one scatter level / two levels
@pytest.mark.parametrize("values", ["k/128", "integers", "halves_negative", "zeros"])
@pytest.mark.parametrize("pred", [True, False])
def test_fixed_point_entries_vs_oracle(groups, values, pred, monkeypatch):
    """Values that are m * 2^qe with few bits travel as 8-byte entry words and are summed as integers: same bits as the oracle, on the
    fused result columns and through finish() (gpu_aggregate compares the two), one level and two levels, two batches."""
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(groups % 1000 + len(values) + int(pred))
    n = 3_000_000
    k = rng.integers(0, groups, n).astype(np.int64) - 17
    if values == "k/128":
        v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    elif values == "integers":
        v = rng.integers(-10**6, 10**6, n).astype(np.float64)
    elif values == "halves_negative":
        v = -rng.integers(0, 5000, n).astype(np.float64) / 2.0 + 100.0
    else:
        v = np.zeros(n)
    t = pa.table({"k": pa.array(k), "v": pa.array(v)})
    batches = util.sliced_batches(t, n // 2)
    predicate = ("v", ">", 1.0 if values != "zeros" else -1.0) if pred else None
    before = _routes()
    got = gpu_aggregate(__import__("oracle.oracle", fromlist=["x"]).SINGLE, ["k"], ["k"], _hot_funcs(), batches, predicate=predicate)
    assert _took(before, "dense:fixed_point") >= 1, "the fixed-point route was not taken"
    assert _took(before, "dense:fixed_point_misfit") == 0
    util.assert_agg_equal(got, _oracle(_hot_funcs(), batches, predicate), _hot_funcs(), ["k"], what=f"fixed point {values} G={groups} pred={pred}")


@pytest.mark.parametrize("misfit", ["finer_fraction", "large", "negative_zero", "nan", "inf", "tenth"])
@pytest.mark.parametrize("where", ["first_batch_late_row", "second_batch"])
def test_fixed_point_misfit_falls_back(misfit, where, monkeypatch):
    """ONE value the sample cannot have announced: a fraction below the quantum, a magnitude beyond 31 bits of quanta, -0.0 (its sign
    would be lost), NaN, Inf, 0.1.  Pass 1 checks every row: the attempt fails, the batch is redone with float64 entries and the
    operator stays there; in the second batch of a stream the first batch's pending fixed-point pass runs first.  Bit-equal to the
    oracle either way (NaN / Inf sums included)."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(len(misfit) * 7 + len(where))
    n, groups = 2_400_000, 1_500_000
    k = rng.integers(0, groups, n).astype(np.int64)
    v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    bad = {"finer_fraction": 3.0 + 2.0**-30, "large": 2.0**40 + 0.5, "negative_zero": -0.0, "nan": float("nan"), "inf": float("inf"), "tenth": 0.1}[misfit]
    half = n // 2
    at = half - 12345 if where == "first_batch_late_row" else n - 777
    # (the value sample reads rows i * rows / 65536 of the first batch: the planted row is not one of them)
    sampled = set(((np.arange(65536, dtype=np.int64) * half) // 65536).tolist())
    while at in sampled:
        at += 1
    v[at] = bad
    t = pa.table({"k": pa.array(k), "v": pa.array(v)})
    batches = util.sliced_batches(t, half)
    funcs = _hot_funcs()
    before = _routes()
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches)
    assert _took(before, "dense:fixed_point_misfit") >= 1, _routes()      # (narrow 32-bit words first: a misfit there tries 64-bit words once)
    if where == "second_batch":
        assert _took(before, "dense:fixed_point") == 1
    util.assert_agg_equal(got, _oracle(funcs, batches), funcs, ["k"], what=f"misfit {misfit} {where}", source=batches)


@pytest.mark.parametrize("groups", [1_500_000, 12_000_000])
def test_narrow_words_fall_back_to_wide_words(groups, monkeypatch):
    """k / 128 values travel as 32-bit words after the last scatter level (|m| < 2^18); ONE value of 1000.5, off the sample's lattice,
    does not fit them but fits the 64-bit words: the attempt is redone with those (one misfit note, the fixed-point route still taken)."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(groups % 313)
    n = 2_400_000
    k = rng.integers(0, groups, n).astype(np.int64)
    v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    half = n // 2
    at = 777_777
    sampled = set(((np.arange(65536, dtype=np.int64) * half) // 65536).tolist())
    while at in sampled:
        at += 1
    v[at] = 1000.5
    t = pa.table({"k": pa.array(k), "v": pa.array(v)})
    batches = util.sliced_batches(t, half)
    funcs = _hot_funcs()
    before = _routes()
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches)
    assert _took(before, "dense:fixed_point_misfit") == 1 and _took(before, "dense:fixed_point") >= 1, _routes()
    util.assert_agg_equal(got, _oracle(funcs, batches), funcs, ["k"], what=f"narrow -> wide words G={groups}")


def test_fixed_point_heavy_group_total_beyond_2e53_quanta(monkeypatch):
    """One group holds 85 % of 1.2e7 rows of values near 2^31 quanta: its total exceeds 2^53 quanta.  Whatever share of it the rings
    spill to the side table (compensated float sums) and whatever share the integer sums of the final pass take, the merged result is
    the correctly rounded exact sum (math.fsum), like the oracle's group count; every other group bit-equal to the oracle."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(11)
    n, groups = 12_000_000, 1_500_000
    k = rng.integers(0, groups, n).astype(np.int64)
    v = rng.integers(2**30, 2**31 - 1, n).astype(np.float64) * 0.25
    v[::1000] = 0.25                           # (the quantum itself is in the sample)
    heavy = rng.random(n) < 0.85
    k[heavy] = 123456
    t = pa.table({"k": pa.array(k), "v": pa.array(v)})
    batches = util.sliced_batches(t, n // 2)
    funcs = _hot_funcs()
    got = util.canon(gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches), ["k"])
    kk = got.column(0).to_numpy()
    i = int(np.searchsorted(kk, 123456))
    exact = math.fsum(v[k == 123456].tolist())
    assert exact > 2.0**53 * 0.25
    assert got.column(1)[i].as_py() == exact, (got.column(1)[i].as_py(), exact)
    assert got.column(3)[i].as_py() == int((k == 123456).sum())
    # (the reference adds in row order and rounds ~1e7 times in that group: the exact-sum bound, not bit equality)
    util.assert_agg_equal(got, _oracle(funcs, batches), funcs, ["k"], exact_float_inputs=(), what="totals beyond 2^53 quanta", source=batches)


@pytest.mark.parametrize("groups", [6_000_000, 12_000_000])          # (two scatter levels: the split final pass of smaller ranges keeps its compensation words)
def test_exact_adds_without_compensation_words(groups, monkeypatch):
    """Values that do NOT fit 31 bits of one quantum (45 significant bits) but whose sums are provably exact (rows * max < 2^53 quanta):
    float64 entries, final pass with {sum, count} slots and non-returning atomics (route dense:exact_adds).  Bit-equal to the oracle."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(groups % 977)
    n = 2_400_000
    k = rng.integers(0, groups, n).astype(np.int64)
    # 33 significant bits of the quantum 2^-6 (|m| up to 2^32: no 31-bit entry word), 2^20 rows: 20 + 32 = 52 <= 53 bits -- provable
    n = 1 << 20
    k = k[:n]
    v = rng.integers(-2**32, 2**32, n).astype(np.float64) * 2.0**-6
    v[5] = (2.0**32 - 1) * 2.0**-6      # (the extremes are there whatever the generator drew)
    v[6] = 2.0**-6
    t = pa.table({"k": pa.array(k), "v": pa.array(v)})
    batches = t.combine_chunks().to_batches()
    funcs = _hot_funcs()
    before = _routes()
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches)
    assert _took(before, "dense:fixed_point") == 0
    assert _took(before, "dense:exact_adds") >= 1, _routes()
    util.assert_agg_equal(got, _oracle(funcs, batches), funcs, ["k"], what=f"exact adds G={groups}", source=batches)


def test_inexact_values_keep_the_compensated_pass(monkeypatch):
    """Lognormal values: neither route applies; the compensated pass gives the correctly rounded exact sum as before."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(3)
    n, groups = 2_000_000, 1_500_000
    k = rng.integers(0, groups, n).astype(np.int64)
    v = rng.lognormal(2.0, 1.0, n)
    t = pa.table({"k": pa.array(k), "v": pa.array(v)})
    batches = util.sliced_batches(t, n // 2)
    funcs = _hot_funcs()
    before = _routes()
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches)
    assert _took(before, "dense:fixed_point") == 0 and _took(before, "dense:exact_adds") == 0
    util.assert_agg_equal(got, _oracle(funcs, batches), funcs, ["k"], exact_float_inputs=(), what="lognormal", source=batches)


@pytest.mark.parametrize("mode", ["stream", "sync"])
def test_fixed_point_stream_of_batches(mode, monkeypatch):
    """A stream of record batches (stream mode: the segments of one launch; synchronous: a pending pass the batches join) with
    fixed-point entries; a ragged last batch; predicate on the value column.  Equal to the oracle."""
    from oracle import oracle as O
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(21)
    n, groups = 3_300_001, 1_500_000
    k = rng.integers(0, groups, n).astype(np.int64)
    v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    t = pa.table({"k": pa.array(k), "v": pa.array(v)})
    batches = util.sliced_batches(t, 1 << 20)
    funcs = _hot_funcs()
    fspec = [(f, 1 if col else None, pa.float64() if col else None) for f, col, _ in funcs]
    before = _routes()
    agg = ops.DeviceAggregate(O.SINGLE, [pa.int64()], fspec, stream_mode=(mode == "stream"))
    agg.set_predicate(">", 64.0)
    keep = []
    for b in batches:
        kc, vc = DeviceColumn.from_arrow(b.column(0)), DeviceColumn.from_arrow(b.column(1))
        keep.append((kc, vc))
        agg.next([kc], [vc, vc, None], pred=vc, nrows=b.num_rows)
    res = agg.result_arrays([0], ["k"], [f[2] for f in funcs])
    agg.close()
    assert _took(before, "dense:fixed_point") >= 1, _routes()
    util.assert_agg_equal(res, _oracle(funcs, batches, ("v", ">", 64.0)), funcs, ["k"], what=f"fixed-point stream ({mode})")


def _ncol_table(rng, n, groups, ncols, misfit_at=None):
    k = rng.integers(0, groups, n).astype(np.int64) + 3
    cols = {"k": pa.array(k)}
    for j, name in enumerate("abc"[:ncols]):
        v = rng.integers(-2**13, 2**13, n).astype(np.float64) / (128.0 if j != 1 else 4.0)
        if misfit_at is not None and j == 1:
            v[misfit_at] = 0.1
        cols[name] = pa.array(v)
    cols["p"] = pa.array(rng.integers(0, 100, n).astype(np.float64))
    return pa.table(cols)


def _ncol_oracle(funcs, batches, pred):
    from oracle import oracle as O
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in batches:
        if pred is not None:
            b = O.filter_batch(b, O.cmp_mask(b.column(b.schema.names.index(pred[0])), O.GT, pred[2]))
        o.next(b)
    return o.result()


@pytest.mark.parametrize("ncols", [2, 3])
@pytest.mark.parametrize("groups", [60_000, 1_500_000, 12_000_000])      # one scatter level / two / two with 256-way fan-outs
@pytest.mark.parametrize("pred", ["input_column", "own_column", "none"])
def test_fixed_point_columns_one_pass_vs_oracle(ncols, groups, pred, monkeypatch):
    """SELECT k, sum(a), avg(b)[, sum(c)], count(b), count(*) ... GROUP BY k over fixed-point-able float64 columns: ONE pass over the rows,
    16-byte entries of two / three values (vnm_agg_fxn.inc).  Two batches (the second one's run is merged with the first's), the
    predicate on an input column, on a column of its own, or none.  Bit-equal to the oracle."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    monkeypatch.setenv("VNM_AGG_SPLIT_DENSE_MIN_ROWS", "500000")
    monkeypatch.setenv("VNM_FXN_MIN_ROWS", "500000")
    rng = np.random.default_rng(groups % 991 + ncols * 13 + len(pred))
    n = 2_400_000
    t = _ncol_table(rng, n, groups, ncols)
    batches = util.sliced_batches(t, n // 2)
    funcs = [(O.SUM, "a", "sa"), (O.AVG, "b", "ab"), (O.COUNT, "b", "cb"), (O.COUNT_STAR, "", "n")]
    if ncols == 3:
        funcs.insert(2, (O.SUM, "c", "sc"))
    predicate = {"input_column": ("a", ">", -20.0), "own_column": ("p", ">", 30.0), "none": None}[pred]
    before = _routes()
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=predicate)
    assert _took(before, "dense:fixed_point_columns") >= 1, _routes()
    util.assert_agg_equal(got, _ncol_oracle(funcs, batches, predicate), funcs, ["k"], exact_float_inputs=("a", "b", "c"), what=f"fixed-point columns C={ncols} G={groups} pred={pred}")


def test_fixed_point_columns_misfit_takes_the_per_column_route(monkeypatch):
    """One 0.1 in the second column, off the sample's lattice: the one-pass attempt fails before anything is written and the batch is cut
    per column as before; equal to the oracle (the float sums of that column within the exact-sum bound)."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    monkeypatch.setenv("VNM_AGG_SPLIT_DENSE_MIN_ROWS", "500000")
    monkeypatch.setenv("VNM_FXN_MIN_ROWS", "500000")
    rng = np.random.default_rng(77)
    n, groups = 2_000_000, 1_500_000
    at = 1_234_567
    sampled = set(((np.arange(65536, dtype=np.int64) * n) // 65536).tolist())
    while at in sampled:
        at += 1
    t = _ncol_table(rng, n, groups, 3, misfit_at=at)
    batches = t.combine_chunks().to_batches()
    funcs = [(O.SUM, "a", "sa"), (O.AVG, "b", "ab"), (O.SUM, "c", "sc"), (O.COUNT_STAR, "", "n")]
    before = _routes()
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=("a", ">", -20.0))
    assert _took(before, "dense:fixed_point_columns_failed") == 1 and _took(before, "dense:fixed_point_columns") == 0, _routes()
    src = [O.filter_batch(b, O.cmp_mask(b.column(1), O.GT, -20.0)) for b in batches]
    util.assert_agg_equal(got, _ncol_oracle(funcs, batches, ("a", ">", -20.0)), funcs, ["k"], exact_float_inputs=("a", "c"), what="misfit in column b", source=src)


def test_fixed_point_columns_stream_segments(monkeypatch):
    """The bench's three-column stream (configs[3] one-GPU leg): recorded batches as the segments of one launch, a ragged last batch."""
    from oracle import oracle as O
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    monkeypatch.setenv("VNM_AGG_SPLIT_DENSE_MIN_ROWS", "500000")
    rng = np.random.default_rng(5)
    n, groups = 3_000_001, 1_000_000
    t = _ncol_table(rng, n, groups, 3)
    batches = util.sliced_batches(t, 1 << 19)
    funcs = [(O.SUM, "a", "sa"), (O.AVG, "b", "ab"), (O.SUM, "c", "sc"), (O.COUNT_STAR, "", "n")]
    fspec = [(O.SUM, 1, pa.float64()), (O.AVG, 2, pa.float64()), (O.SUM, 3, pa.float64()), (O.COUNT_STAR, None, None)]
    before = _routes()
    agg = ops.DeviceAggregate(O.SINGLE, [pa.int64()], fspec, stream_mode=True)
    agg.set_predicate(">", -20.0)
    keep = []
    for b in batches:
        cols = [DeviceColumn.from_arrow(b.column(i)) for i in range(4)]
        keep.append(cols)
        agg.next([cols[0]], [cols[1], cols[2], cols[3], None], pred=cols[1], nrows=b.num_rows)
    res = agg.result_arrays([0], ["k"], [f[2] for f in funcs])
    agg.close()
    assert _took(before, "dense:fixed_point_columns") >= 1, _routes()
    util.assert_agg_equal(res, _ncol_oracle(funcs, batches, ("a", ">", -20.0)), funcs, ["k"], exact_float_inputs=("a", "b", "c"), what="fixed-point columns, stream")


def test_parquet_and_json_sources_through_the_pinned_ring(tmp_path):
    """vinum.read_parquet / read_json (io/arrow.py:111-248) as streams of device record batches: pyarrow decodes, every batch is staged
    through vnm_stage_column (numeric columns, validity with an odd offset) or the device dictionary (strings); GROUP BY over the
    stream == pyarrow's group_by over the file; column pruning reaches the reader."""
    import pyarrow.parquet as pq
    from vinum_amd import planner
    from vinum_amd.io import stream_parquet, read_parquet, read_json
    rng = np.random.default_rng(9)
    n = 250_003
    t = pa.table({"id": pa.array(np.arange(n, dtype=np.int64)),
                  "k": pa.array(rng.integers(0, 97, n).astype(np.int32), mask=rng.random(n) < 0.01),
                  "v": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0, mask=rng.random(n) < 0.05),
                  "city": pa.array(rng.choice(["Berlin", "Riva", "München", "Malaga", ""], n))})
    path = os.path.join(tmp_path, "t.parquet")
    pq.write_table(t, path, row_group_size=60_000)
    q = dict(select=["k", ["fn", "count_star"], ["fn", "sum", "v"], ["fn", "count", "v"]], aliases=[None, "n", "s", "c"], group_by=["k"])
    got = planner.execute(q, stream_parquet(path, batch_rows=33_333)).sort_by("k")
    exp = t.group_by("k", use_threads=False).aggregate([([], "count_all"), ("v", "sum"), ("v", "count")]).sort_by("k")
    assert got.column("k").to_pylist() == exp.column("k").to_pylist()
    assert got.column("n").cast(pa.int64()).to_pylist() == exp.column("count_all").to_pylist()
    assert got.column("c").cast(pa.int64()).to_pylist() == exp.column("v_count").to_pylist()
    assert got.column("s").to_pylist() == exp.column("v_sum").to_pylist()          # (quantised values: every sum is exact)
    # a string key through the device dictionary, the whole file as one batch
    q2 = dict(select=["city", ["fn", "count_star"]], aliases=[None, "n"], group_by=["city"])
    got2 = planner.execute(q2, stream_parquet(path, batch_rows=100_000)).sort_by("city")
    exp2 = t.group_by("city", use_threads=False).aggregate([([], "count_all")]).sort_by("city")
    assert got2.column("city").to_pylist() == exp2.column("city").to_pylist() and got2.column("n").cast(pa.int64()).to_pylist() == exp2.column("count_all").to_pylist()
    whole = read_parquet(path, columns=["id", "v"])
    assert whole.num_rows == n and whole.columns["v"].to_arrow().equals(t.column("v").combine_chunks())
    # line-delimited JSON
    jpath = os.path.join(tmp_path, "t.json")
    small = t.slice(0, 5000)
    with open(jpath, "w") as f:
        for row in small.to_pylist():
            f.write(__import__("json").dumps(row) + "\n")
    rows = 0
    for b in read_json(jpath, batch_rows=2000):
        rows += b.num_rows
        assert b.columns["id"].to_arrow().type == pa.int64()
    assert rows == 5000


def _raw_aggregate(kind, groupby, agg_cols, funcs, batches) -> pa.RecordBatch:
    """vnm_agg_op_create / _next / _result through ctypes and the Arrow C Data Interface only -- what a binding of include/vinum_hip.h
    in any language does (vinum/core/vinum_lib.cpp:54-124).  No vinum_lib.py between the test and the library."""
    from vinum_amd import _lib as L
    lib = L.lib()
    cs = lambda xs: (ctypes.c_char_p * max(len(xs), 1))(*[x.encode() for x in xs])
    ft = (ctypes.c_int * len(funcs))(*[int(f[0]) for f in funcs])
    h = lib.vnm_agg_op_create(int(kind), len(groupby), cs(groupby), len(agg_cols), cs(agg_cols), len(funcs), ft, cs([f[1] for f in funcs]), cs([f[2] for f in funcs]))
    assert h, L.last_error()
    try:
        for b in batches:
            arr, sch = ctypes.create_string_buffer(80), ctypes.create_string_buffer(72)
            b._export_to_c(ctypes.addressof(arr), ctypes.addressof(sch))
            L.check(lib.vnm_agg_op_next(h, ctypes.addressof(arr), ctypes.addressof(sch)))
        arr, sch = ctypes.create_string_buffer(80), ctypes.create_string_buffer(72)
        L.check(lib.vnm_agg_op_result(h, ctypes.addressof(arr), ctypes.addressof(sch)))
        return pa.RecordBatch._import_from_c(ctypes.addressof(arr), ctypes.addressof(sch))
    finally:
        lib.vnm_agg_op_destroy(h)


@pytest.mark.parametrize("seed", range(12))
def test_generic_keys_through_the_raw_c_abi(seed):
    """GenericHashAggregate's keys (generic_hash_aggregate.h:10-45; bound at vinum_lib.cpp:92-109) below the C ABI: utf8 / large_utf8 /
    binary / boolean / decimal128 group keys -- alone, mixed with an int64 key -- with NULLs, empty strings and values first seen in a later
    batch (small batches wait and are staged together, a large one goes alone), COUNT over a string column next to numeric functions:
    raw ctypes calls on vnm_agg_op_*, equal to the oracle's restatement (OracleGenericAggregate) group for group, types included."""
    import decimal
    from oracle import oracle as O
    rng = np.random.default_rng(4200 + seed)
    n = int(rng.integers(4000, 15000))
    words = ["", "a", "A", "ab", "Berlin", "Munich", "San Francisco", "zürich", "0", "00", "été"] + [f"w{int(x)}" for x in rng.integers(0, 300, 50)]
    kt = [pa.string(), pa.large_string(), pa.binary(), pa.bool_(), pa.decimal128(12, 2), pa.string()][seed % 6]
    def key_col():
        if pa.types.is_boolean(kt):
            return pa.array([None if rng.random() < 0.1 else bool(x) for x in rng.integers(0, 2, n)], type=kt)
        if pa.types.is_decimal(kt):
            return pa.array([None if rng.random() < 0.05 else decimal.Decimal(int(x)) / 100 for x in rng.integers(-500, 500, n)], type=kt)
        v = [None if rng.random() < 0.05 else str(x) for x in rng.choice(words, n)]
        return pa.array([x.encode() if (x is not None and pa.types.is_binary(kt)) else x for x in v], type=kt)
    t = pa.table({"g": key_col(), "k": pa.array(rng.integers(-5, 5, n).astype(np.int64), mask=rng.random(n) < 0.05),
                  "s": pa.array([None if rng.random() < 0.2 else str(x) for x in rng.choice(words, n)], type=pa.string()),
                  "v": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0, mask=rng.random(n) < 0.1)})
    two_keys = seed >= 6
    groupby = ["g", "k"] if two_keys else ["g"]
    funcs = [(O.COUNT_STAR, "", "n"), (O.COUNT, "s", "cs"), (O.SUM, "v", "sv"), (O.MAX, "v", "xv"), (O.COUNT, "g", "cg")]
    cuts = sorted(set(int(x) for x in rng.integers(1, n - 1, 3)))
    bounds = [0] + cuts + [n]
    batches = [t.slice(a, b - a).combine_chunks().to_batches()[0] for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    kind = O.MULTI if two_keys else O.SINGLE
    got = _raw_aggregate(kind, groupby, groupby, funcs, batches)
    o = O.OracleGenericAggregate(3, groupby, groupby, funcs)
    for b in batches:
        o.next(b)
    exp = o.result()
    assert got.schema.names == exp.schema.names and [f.type for f in got.schema] == [f.type for f in exp.schema], (got.schema, exp.schema)
    def keyed(batch):
        rows = list(zip(*[batch.column(i).to_pylist() for i in range(batch.num_columns)]))
        return sorted(rows, key=lambda r: tuple((x is None, x if x is not None else 0) for x in r[:len(groupby)]))
    g_rows, e_rows = keyed(got), keyed(exp)
    assert len(g_rows) == len(e_rows), f"{len(g_rows)} groups vs {len(e_rows)}"
    assert g_rows == e_rows, f"first difference {[(a, b) for a, b in zip(g_rows, e_rows) if a != b][:2]}"


def test_string_min_max_over_a_long_stream_keeps_a_bounded_number_of_partials():
    """string MIN / MAX kept one partial result (groups x functions, as host strings) per input batch until
    result().  300 batches: the partials are folded every eight batches; same answer as pyarrow's group_by."""
    from vinum_amd import vinum_lib as vl
    rng = np.random.default_rng(12)
    words = [f"w{int(x):04d}" for x in range(500)] + ["", "Zebra", "äpfel"]
    agg = vl.SingleNumericalHashAggregate(["k"], ["k"], [vl.AggFuncDef(vl.MIN, "s", "mn"), vl.AggFuncDef(vl.MAX, "s", "mx"), vl.AggFuncDef(vl.COUNT_STAR, "", "n")])
    agg._SMALL_ROWS = 1            # (every batch crosses the boundary on its own: no coalescing in the wrapper)
    tables = []
    for i in range(300):
        n = 2000
        t = pa.table({"k": pa.array(rng.integers(0, 40, n).astype(np.int64)),
                      "s": pa.array([None if rng.random() < 0.1 else str(x) for x in rng.choice(words, n)], type=pa.string())})
        tables.append(t)
        for b in t.to_batches():
            agg.next(b)
        if agg._strmm is not None:
            assert len(agg._strmm._partials) <= 8
    got = agg.result().to_pydict()
    exp = pa.concat_tables(tables).group_by("k", use_threads=False).aggregate([("s", "min"), ("s", "max"), ([], "count_all")]).to_pydict()
    g = {k: (a, b, int(c)) for k, a, b, c in zip(got["k"], got["mn"], got["mx"], got["n"])}
    e = {k: (a.encode() if a is not None else None, b.encode() if b is not None else None, c) for k, a, b, c in zip(exp["k"], exp["s_min"], exp["s_max"], exp["count_all"])}
    # (byte-wise order, as StringMinMaxFunc compares string_views: agg_funcs.h:219-261)
    import collections
    by = collections.defaultdict(list)
    for t in tables:
        for k, s_ in zip(t.column("k").to_pylist(), t.column("s").to_pylist()):
            if s_ is not None:
                by[k].append(s_.encode())
    for k in e:
        mn, mx = (min(by[k]), max(by[k])) if by[k] else (None, None)
        ga = g[k]
        assert (ga[0].encode() if ga[0] is not None else None, ga[1].encode() if ga[1] is not None else None, ga[2]) == (mn, mx, e[k][2]), (k, ga, mn, mx)


def test_float32_predicate_literal_after_the_ordered_switch():
    """`WHERE v > 0.1` over a float32 column compares in float32 (NumPy: float32 array against a Python scalar): float32(0.1)
    is NOT above the literal.  The column travels widened and the literal rounded; an operator that has switched to the ordered
    float MIN / MAX mode (a NaN in batch 2) feeds its suffix operator the widened column too -- a set_predicate after the switch must reach
    it as the same rounded number."""
    from oracle import oracle as O
    from vinum_amd.device import DeviceColumn
    from vinum_amd import ops
    rng = np.random.default_rng(8)
    tenth = np.float32(0.1)
    def batch(n, with_nan):
        v = rng.choice(np.array([0.05, 0.1, 0.25, 3.0, -1.0], np.float32), n)
        if with_nan:
            v[::97] = np.nan
        return pa.RecordBatch.from_pydict({"k": pa.array(rng.integers(0, 50, n).astype(np.int64)), "v": pa.array(v)})
    batches = [batch(40_000, False), batch(40_000, True), batch(40_000, False)]
    funcs = [(O.MIN, "v", "lo"), (O.MAX, "v", "hi"), (O.COUNT, "v", "c"), (O.SUM, "v", "s")]
    fspec = [(f, 1, pa.float32()) for f, _, _ in funcs]
    agg = ops.DeviceAggregate(O.SINGLE, [pa.int64()], fspec)
    agg.set_predicate(">", 0.1)
    for i, b in enumerate(batches):
        if i == 2:
            agg.set_predicate(">", 0.1)        # (again, after the switch: the suffix operator exists by now)
        kc, vc = DeviceColumn.from_arrow(b.column(0)), DeviceColumn.from_arrow(b.column(1))
        agg.next([kc], [vc] * 4, pred=vc, nrows=b.num_rows)
    got = agg.result_arrays([0], ["k"], [f[2] for f in funcs])
    agg.close()
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in batches:
        v = b.column(1).to_numpy(zero_copy_only=False)
        with np.errstate(invalid="ignore"):
            o.next(O.filter_batch(b, np.asarray(v > tenth)))   # float32 comparison (NaN > x is False)
    util.assert_agg_equal(got, o.result(), funcs, ["k"], exact_float_inputs=("v",), what="float32 predicate literal after the switch")


class _RawI64w:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}


@pytest.mark.parametrize("case", ["f64_uniform", "f64_normal_desc", "f64_lognormal", "i64_wide", "u64_desc", "few_duplicates", "nan_and_zeros", "outside_the_sample",
                                  "duplicates_decline", "lumpy_declines", "fanout_64", "fanout_512", "fanout_16_native", "odd_unaligned", "f32_widened",
                                  "nulls_asc", "nulls_desc_unaligned", "mostly_null_declines", "sorted_input_declines", "sorted_runs_decline", "already_sorted", "already_sorted_desc_with_ties", "already_sorted_backwards"])
def test_entry_word_sample_sort_equals_the_lsd_sort(case, monkeypatch):
    """vnm_sort_indices over one 8-byte key when only the order is asked for (vnm_sort_apx.inc): rows travel as 8-byte words
    (a32 << 32 | row id) where a32 comes from an equalising piecewise-linear map of the code; two entries with the same a32 are
    ordered through the key column.  The row ids must equal the eight-pass LSD sort's (the order is total: key, then row id --
    Sort::Sorted is stable, sort.cpp:22-40).  Data with duplicated keys or a lumpy distribution must decline (and still be right)."""
    import torch
    from vinum_amd import _lib as L
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    rng = np.random.default_rng(4242)
    n = 3_000_000
    order = L.ASC
    taken = True
    if case == "f64_uniform":
        v = rng.random(n)
    elif case == "f64_normal_desc":
        v = rng.normal(0.0, 1.0, n); order = L.DESC
    elif case == "f64_lognormal":
        v = np.exp(rng.normal(0.0, 6.0, n))
    elif case == "i64_wide":
        v = rng.integers(-2**62, 2**62, n).astype(np.int64)
    elif case == "u64_desc":
        v = rng.integers(0, 2**63, n).astype(np.uint64) * 2 + rng.integers(0, 2, n).astype(np.uint64); order = L.DESC
    elif case == "few_duplicates":                     # 1 % of the rows repeat another row's key: ordered through the key column
        v = rng.normal(0.0, 1.0, n); src = rng.integers(0, n, n // 100); v[rng.integers(0, n, n // 100)] = v[src]
    elif case in ("nan_and_zeros", "outside_the_sample"):
        # rows the strided sample (row i * n / m) does not see: NaN / -0.0 / +0.0, values below and above everything sampled
        m = 65536
        monkeypatch.setenv("VNM_XSORT_SAMPLE", str(m))
        sampled = np.zeros(n, dtype=bool); sampled[(np.arange(m, dtype=np.int64) * n) // m] = True
        free = np.nonzero(~sampled)[0]
        pick = free[rng.choice(len(free), 200, replace=False)]
        v = rng.normal(0.0, 1.0, n)
        if case == "nan_and_zeros":
            v[pick[:60]] = np.nan; v[pick[60:120]] = -0.0; v[pick[120:180]] = 0.0
        else:
            v[pick[:80]] = -1e300; v[pick[80:160]] = 1e300; v[pick[160]] = -np.inf; v[pick[161]] = np.inf; v[pick[162:200]] = 1e-300
    elif case == "duplicates_decline":
        v = rng.integers(0, n // 4, n).astype(np.float64); taken = False
    elif case == "lumpy_declines":                     # tight clusters: not linear inside a cell
        v = rng.integers(0, 50, n).astype(np.float64) * 1000.0 + rng.random(n) * 1e-6; taken = False
    elif case in ("fanout_64", "fanout_512"):
        monkeypatch.setenv("VNM_XSORT_L2", case.split("_")[1]); v = rng.normal(5.0, 2.0, n)
    elif case == "fanout_16_native":
        n = (1 << 25) + 12345; v = rng.random(n) * 1e6 - 5e5; order = L.DESC
    elif case == "odd_unaligned":
        n = 2_999_999 + 4096 * 3 + 17; v = rng.normal(0.0, 1.0, n)
    elif case == "f32_widened":
        n = (1 << 21) + 77; v = rng.permutation(np.arange(n, dtype=np.int64) * 2000 - 2**31 + 5).astype(np.int32)    # distinct int32 keys
    elif case == "sorted_input_declines":              # rows that arrive in key order: every sub-tile would go to one ring -- both sample sorts decline
        v = np.sort(rng.normal(0.0, 1.0, n)); v[100] = v[99]; order = L.DESC; taken = False      # (a tie: not simply the row numbers backwards)
    elif case == "already_sorted":                     # ... and when that order IS the order asked for, the indices are the row numbers
        v = np.sort(rng.normal(0.0, 1.0, n)); v[-3:] = np.nan; taken = False
    elif case == "already_sorted_desc_with_ties":
        v = -np.sort(rng.integers(0, n // 3, n).astype(np.float64)); v[v == 0.0] = 0.0; v[5] = v[4]; order = L.DESC; taken = False
    elif case == "already_sorted_backwards":           # strictly ascending, asked for descending, no ties: the row numbers backwards
        v = np.sort(rng.permutation(n).astype(np.float64) * 0.5); order = L.DESC; taken = False
    elif case == "sorted_runs_decline":                # ... or in sorted runs of 2^16 rows
        v = rng.normal(0.0, 1.0, n); v = np.concatenate([np.sort(v[i:i + 65536]) for i in range(0, n, 65536)]); taken = False
    elif case in ("nulls_asc", "nulls_desc_unaligned", "mostly_null_declines"):
        v = rng.normal(0.0, 1.0, n)
        order = L.DESC if "desc" in case else L.ASC
        taken = case != "mostly_null_declines"
    else:
        raise AssertionError(case)
    if case.startswith("nulls") or case == "mostly_null_declines":
        # NULL keys: their rows stand behind every value in both directions, in row order (SortIndices' null placement, sort.cpp:22-37)
        mask = rng.random(n) < (0.02 if case == "nulls_asc" else (0.3 if case != "mostly_null_declines" else 0.8))
        arr = pa.array(v, mask=mask)
        if "unaligned" in case:
            arr = arr.slice(5, n - 11); n = len(arr)
        col = DeviceColumn.from_arrow(arr)
    elif case == "odd_unaligned":
        arr = pa.array(np.concatenate([[0.0] * 3, v])).slice(3, n)      # an odd Arrow offset: no 16-byte pairs
        col = DeviceColumn.from_arrow(arr)
    else:
        t = torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).cuda()
        col = DeviceColumn.from_torch(t)
        if v.dtype == np.uint64:
            col = DeviceColumn(col._values, None, 0, n, pa.uint64(), keep=t)
    monkeypatch.setenv("VNM_SORT_NO_SAMPLE", "1")
    ref_idx = ops.sort_indices([col], [order])
    ref = torch.as_tensor(_RawI64w(ref_idx.ptr, n), device="cuda").clone()
    monkeypatch.delenv("VNM_SORT_NO_SAMPLE")
    monkeypatch.setenv("VNM_SSORT_MIN_ROWS", "1000")
    before = _routes()
    got_idx = ops.sort_indices([col], [order])
    assert (_took(before, "sort:sample_sort_words") >= 1) == taken, (case, _took(before, "sort:sample_sort_words"))
    if case.startswith("sorted_"):
        assert _took(before, "sort:lsd_radix") >= 1, "rows in key order must go to the LSD passes"
    got = torch.as_tensor(_RawI64w(got_idx.ptr, n), device="cuda")
    if case.startswith("already_sorted"):
        assert _took(before, "sort:already_sorted") >= 1 and _took(before, "sort:lsd_radix") == 0
        assert bool(torch.equal(got, torch.arange(n, device="cuda") if case != "already_sorted_backwards" else torch.arange(n - 1, -1, -1, device="cuda")))
    assert bool(torch.equal(got, ref)), f"{case}: {int((got != ref).sum())} positions differ"


@pytest.mark.parametrize("groups", [1_500_000, 12_000_000])
@pytest.mark.parametrize("what", ["null_keys", "null_keys_pred_on_other", "null_values", "null_values_misfit"])
def test_fixed_point_entries_with_nulls_vs_oracle(what, groups, monkeypatch):
    """Fixed-point entry words with a NULLABLE key (pass 1 sums the NULL-key rows aside, dring_scatter_kernel<..., KN, ..., FX>) and with a
    nullable VALUE column that the query's own filter reads (vn_fold: NULL > x is not true, the rows never become entries).  Equal to the oracle;
    a misfit among the surviving values still redoes the batch with float64 entries."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(groups % 911 + len(what))
    n = 2_400_000
    k = rng.integers(0, groups, n).astype(np.int64)
    v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    kmask = vmask = None
    cols = {}
    if what.startswith("null_keys"):
        kmask = rng.random(n) < 0.12
    else:
        vmask = rng.random(n) < 0.05
        if what == "null_values_misfit":
            v[n // 2 + 7] = 100.0 + 2.0**-30        # (a surviving value that is no multiple of the quantum)
            v[5] = 1e-300; vmask[5] = True          # (a NULL slot with bits that would not fit: never looked at)
    cols["k"] = pa.array(k, mask=kmask)
    cols["v"] = pa.array(v, mask=vmask)
    pred_col = "v"
    if what == "null_keys_pred_on_other":
        cols["p"] = pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)
        pred_col = "p"
    t = pa.table(cols)
    batches = t.combine_chunks().to_batches()
    predicate = (pred_col, ">", 64.0)
    funcs = _hot_funcs()
    before = _routes()
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=predicate)
    if what == "null_values_misfit":
        assert _took(before, "dense:fixed_point_misfit") >= 1
    else:
        assert _took(before, "dense:fixed_point") >= 1, _routes()
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in batches:
        o.next(O.filter_batch(b, O.cmp_mask(b.column(b.schema.names.index(pred_col)), O.GT, 64.0)))
    util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"fixed point + {what} G={groups}")


@pytest.mark.parametrize("seed", range(12))
def test_string_min_max_through_the_raw_c_abi(seed):
    """StringMinMaxFunc (agg_funcs.h:219-261; Single_/Multi_/Generic_Int64Grp_StringArgFuncs, hash_agg_test.cpp:866-894) below the C ABI:
    MIN / MAX over utf8 / large_utf8 / binary / large_binary columns next to numeric functions and COUNT over the same column, with an
    int64 key, a string key, two keys, no key at all; NULLs, all-NULL groups, empty strings, values first seen in later batches, small
    batches that wait and a large one that goes alone.  Raw ctypes calls on vnm_agg_op_*; equal to the oracle's row loop, types included."""
    from oracle import oracle as O
    rng = np.random.default_rng(5200 + seed)
    n = int(rng.integers(4000, 15000))
    words = ["", "a", "A", "ab", "abc", "Berlin", "Munich", "San Francisco", "zürich", "0", "00", "été", "a\x00b", "a\x00"] + [f"w{int(x)}" for x in rng.integers(0, 300, 50)]
    st = [pa.string(), pa.large_string(), pa.binary(), pa.large_binary()][seed % 4]
    def str_col(null_p, typ=st, pool=words):
        v = [None if rng.random() < null_p else str(x) for x in rng.choice(pool, n)]
        return pa.array([x.encode() if (x is not None and (pa.types.is_binary(typ) or pa.types.is_large_binary(typ))) else x for x in v], type=typ)
    k = rng.integers(-20, 20, n).astype(np.int64)
    s = str_col(0.2).to_pylist()
    for i in range(n):                      # (some groups see NULLs only)
        if k[i] in (-7, 3):
            s[i] = None
    t = pa.table({"k": pa.array(k, mask=rng.random(n) < 0.05), "g": str_col(0.05, pa.string(), ["x", "y", "zz", "", "Ω"]),
                  "s": pa.array(s, type=st), "s2": str_col(0.5, pa.string()),
                  "v": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0, mask=rng.random(n) < 0.1)})
    shape = ["int_key", "string_key", "two_keys", "one_group", "strings_only", "int_key"][seed % 6]
    groupby = {"int_key": ["k"], "string_key": ["g"], "two_keys": ["g", "k"], "one_group": [], "strings_only": ["k"]}[shape]
    funcs = [(O.MIN, "s", "mn"), (O.COUNT_STAR, "", "n"), (O.MAX, "s", "mx"), (O.COUNT, "s", "cs"), (O.SUM, "v", "sv"), (O.MIN, "s2", "mn2"), (O.MAX, "v", "xv")]
    if shape == "strings_only":
        funcs = [(O.MAX, "s", "mx"), (O.MIN, "s", "mn")]
    cuts = sorted(set(int(x) for x in rng.integers(1, n - 1, 3)))
    bounds = [0] + cuts + [n]
    batches = [t.slice(a, b - a).combine_chunks().to_batches()[0] for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    kind = {0: O.ONE_GROUP, 1: O.SINGLE, 2: O.MULTI}[len(groupby)]
    got = _raw_aggregate(kind, groupby, groupby, funcs, batches)
    o = O.OracleGenericAggregate(3 if groupby else O.ONE_GROUP, groupby, groupby, funcs)
    for b in batches:
        o.next(b)
    exp = o.result()
    assert got.schema.names == exp.schema.names and [f.type for f in got.schema] == [f.type for f in exp.schema], (got.schema, exp.schema)
    def keyed(batch):
        rows = list(zip(*[batch.column(i).to_pylist() for i in range(batch.num_columns)]))
        return sorted(rows, key=lambda r: tuple((x is None, x if x is not None else 0) for x in r[:len(groupby)]))
    g_rows, e_rows = keyed(got), keyed(exp)
    assert len(g_rows) == len(e_rows), f"{len(g_rows)} groups vs {len(e_rows)}"
    assert g_rows == e_rows, f"first difference {[(a, b) for a, b in zip(g_rows, e_rows) if a != b][:2]}"


def test_string_min_max_below_the_abi_folds_its_candidates():
    """Many groups and several device-sized batches: every batch leaves one candidate per group and function in HBM, and the chunks are
    folded when they hold more than twice the groups -- the candidates' ids are ranked against the dictionary of THAT moment (values that
    arrive later sort between earlier ones).  Equal to pyarrow's group_by over the binary column (byte-wise order)."""
    rng = np.random.default_rng(99)
    groups, nb, rows = 90_000, 6, (1 << 20) + 4096
    tables = []
    for i in range(nb):
        k = rng.integers(0, groups, rows).astype(np.int64)
        # later batches bring values that sort BEFORE and BETWEEN the earlier ones
        x = rng.integers(0, 400_000, rows) * (nb - i)
        sv = pa.array(np.char.add("v", np.char.zfill(x.astype(str), 8)), mask=rng.random(rows) < 0.1)
        tables.append(pa.table({"k": pa.array(k), "s": sv.cast(pa.binary())}))
    from oracle import oracle as O
    funcs = [(O.MIN, "s", "mn"), (O.MAX, "s", "mx"), (O.COUNT_STAR, "", "n")]
    got = _raw_aggregate(O.SINGLE, ["k"], ["k"], funcs, [t.combine_chunks().to_batches()[0] for t in tables])
    exp = pa.concat_tables(tables).group_by("k", use_threads=False).aggregate([("s", "min"), ("s", "max"), ([], "count_all")])
    g = pa.Table.from_batches([got]).sort_by("k")
    e = exp.sort_by("k")
    assert g.num_rows == e.num_rows == len(set(np.concatenate([t.column("k").to_numpy() for t in tables]).tolist()))
    assert g.column("mn").combine_chunks().equals(e.column("s_min").combine_chunks())
    assert g.column("mx").combine_chunks().equals(e.column("s_max").combine_chunks())
    assert g.column("n").to_pylist() == e.column("count_all").to_pylist()


@pytest.mark.parametrize("case", ["one_level_100k", "two_levels_2m", "pred_on_other", "heavy_key_falls_back"])
def test_sparse_keys_through_the_ring_form_of_the_hash_partitions(case, monkeypatch):
    """Keys that are no dense range (random 62-bit values): the hash partitions.  Where a level has 128 .. 256 partitions its (key, value)
    entries go through LDS rings (pring_scatter_kernel); a heavy key fails that attempt and the tile-sorting scatter redoes the batch.  Equal to
    the oracle either way."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(len(case))
    groups = 100_000 if case == "one_level_100k" else 2_000_000
    n = 3_000_000
    vals = rng.integers(-2**62, 2**62, groups).astype(np.int64)
    k = vals[rng.integers(0, groups, n)]
    if case == "heavy_key_falls_back":
        k[rng.random(n) < 0.3] = vals[0]
    v = rng.integers(0, 2**14, n).astype(np.float64) / 128.0
    cols = {"k": pa.array(k), "v": pa.array(v)}
    pred_col = "v"
    if case == "pred_on_other":
        cols["p"] = pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0); pred_col = "p"
    t = pa.table(cols)
    batches = util.sliced_batches(t, n // 2)
    funcs = _hot_funcs()
    before = _routes()
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=(pred_col, ">", 64.0), expected_groups=groups)
    assert _took(before, "hash_partitions:hot") >= 1, _routes()
    assert _took(before, "hash_partitions:rings") + _took(before, "hash_partitions:rings_failed") >= 1, _routes()
    if case == "heavy_key_falls_back":
        assert _took(before, "hash_partitions:rings_failed") >= 1, _routes()
    o = O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    for b in batches:
        o.next(O.filter_batch(b, O.cmp_mask(b.column(b.schema.names.index(pred_col)), O.GT, 64.0)))
    util.assert_agg_equal(got, o.result(), funcs, ["k"], what=f"sparse keys {case}")
