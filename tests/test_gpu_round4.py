"""GPU parity, round 4: operator-state transitions the earlier suites never crossed -- NULL keys that first
appear in a LATER batch of a stream whose first batches took the dense / partitioned / split / stream-table routes --
and the asynchronous stream mode (no host read-back per next())."""
import ctypes
import os

import numpy as np
import pyarrow as pa
import pytest

from tests import util
from tests.test_gpu_agg import gpu_aggregate

pytestmark = pytest.mark.gpu


def _oracle(kind, keys, funcs, batches, pred=None):
    from oracle import oracle as O
    o = O.OracleAggregate(kind, keys, keys, funcs)
    for b in batches:
        if pred:
            b = O.filter_batch(b, O.cmp_mask(b.column(b.schema.names.index(pred[0])), O.GT, pred[2]))
        o.next(b)
    return o.result()


@pytest.mark.parametrize("later", ["null_keys", "odd_offset", "null_keys_and_key_zero"])
@pytest.mark.parametrize("route", ["dense_two_level", "dense_one_level", "dense_split_final", "stream_table", "hash_partitions",
                                   "split_program"])
@pytest.mark.parametrize("hint", [0, 1])
def test_null_keys_first_appear_in_a_later_batch(route, later, hint, monkeypatch):
    """ADVICE r03 (high + medium): the first batches of a single 8-byte key come without a validity bitmap and take a plain-key
    route (dense deferred pass, hash partitions, the stream table of the small-range scan, the parts of a split program); a later
    batch brings NULL keys (or an odd Arrow offset) and is not `key_plain` any more.  The operator used to enter packed mode
    there and to return the inner operator's groups ONLY; the parts of a split program turned the NULL group into key 0.
    base_aggregate.cpp:23-45 (state is batch-commutative), single_numerical_hash_aggregate.cpp:24-32 (null_group)."""
    from oracle import oracle as O
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    groups = {"dense_two_level": 1_500_000, "dense_one_level": 900_000, "dense_split_final": 40_000, "stream_table": 3_000,
              "hash_partitions": 200_000, "split_program": 40_000}[route]
    if route == "dense_two_level":
        monkeypatch.setenv("VNM_DENSE_ONE_LEVEL", "0")
    rng = np.random.default_rng(len(route) * 7 + len(later) + hint)
    n1, n2 = (1_200_000, 700_001) if route != "split_program" else (400_000, 233_335)   # (eight columns: the exact-sum check is the slow part)
    mult = 7919 if route == "hash_partitions" else 1     # sparse keys: no dense code range

    def keys(n):
        return rng.integers(0, groups, n).astype(np.int64) * mult

    ncols = 8 if route == "split_program" else 1
    kinds = [O.SUM, O.MAX, O.AVG, O.MIN, O.COUNT, O.SUM, O.AVG, O.MAX]

    def table(n, nullable):
        k = keys(n)
        k[::1013] = 0      # the real key 0 next to the NULL group (whose key word is 0 as well)
        cols = {}
        if nullable:
            mask = rng.random(n) < 0.07
            if later == "null_keys_and_key_zero":
                mask[k == 0] = (np.arange(int((k == 0).sum())) % 2 == 0)
            cols["k"] = pa.array(k, mask=mask)
        else:
            cols["k"] = pa.array(k)
        for c in range(ncols):
            vals = rng.integers(0, 2**14, n).astype(np.float64) / 64.0 if c % 2 == 0 else rng.integers(-2**40, 2**40, n).astype(np.int64)
            cols["v" if ncols == 1 else f"c{c}"] = pa.array(vals)
        return pa.table(cols)

    t1, t2, t3 = table(n1, False), table(n2, later != "odd_offset"), table(n1 // 2, False)
    b2 = t2.to_batches()[0]
    if later == "odd_offset":
        b2 = b2.slice(1)
    batches = t1.to_batches() + [b2] + t3.to_batches()
    if ncols == 1:
        funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT_STAR, "", "n")]
        pred = ("v", ">", 64.0)
    else:
        funcs = [(kinds[c], f"c{c}", f"f{c}") for c in range(ncols)] + [(O.COUNT_STAR, "", "n")]
        pred = None
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=pred, expected_groups=groups if hint else 0)
    exp = _oracle(O.SINGLE, ["k"], funcs, batches, pred)
    assert got.num_rows == exp.num_rows, (got.num_rows, exp.num_rows)
    util.assert_agg_equal(got, exp, funcs, ["k"], what=f"{route}, later batch with {later}", source=pa.Table.from_batches(batches) if not pred else None)


def _launches(name):
    from vinum_amd import _lib as L
    ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
    L.lib().vnm_profile_query(name, ctypes.byref(ms), ctypes.byref(cnt))
    return cnt.value


@pytest.mark.parametrize("pred", ["on_input", "on_other", "none"])
@pytest.mark.parametrize("route", ["dense_two_level", "dense_one_level", "dense_split_final", "lds_scan_g7", "stream_table_g3000",
                                   "hash_partitions", "shape_changes_midstream", "generic_program"])
def test_async_stream_of_batches(route, pred, monkeypatch):
    """vnm_agg_set_async: the batches of a stream wait in the operator and go to the device as the SEGMENTS of one launch (dense ring
    scatter, hot LDS scan) or one by one (every other path).  Same result as the oracle fed batch by batch
    (base_aggregate.cpp:23-45: the state does not depend on where the batches are cut) -- ragged batch sizes (odd, tiny, empty, not a
    multiple of any tile), a batch of another shape in the middle of the stream, a second sync point."""
    from oracle import oracle as O
    from vinum_amd import _lib as L, ops
    from vinum_amd.device import DeviceColumn
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    groups = {"dense_two_level": 1_500_000, "dense_one_level": 900_000, "dense_split_final": 40_000, "lds_scan_g7": 7,
              "stream_table_g3000": 3_000, "hash_partitions": 200_000, "shape_changes_midstream": 900_000, "generic_program": 900_000}[route]
    if route == "dense_two_level":
        monkeypatch.setenv("VNM_DENSE_ONE_LEVEL", "0")
    mult = 7919 if route == "hash_partitions" else 1
    rng = np.random.default_rng(len(route) * 13 + len(pred))
    sizes = [600_000, 262_144, 300_001, 77, 8192, 123_457, 499_999, 16_384 + 2, 650_000, 0]   # (sync after the seventh; the empty batch flushes too)

    batches = []
    for i, n in enumerate(sizes):
        if n == 0:
            batches.append(pa.RecordBatch.from_pydict({"k": pa.array([], pa.int64()), "v": pa.array([], pa.float64()), "p": pa.array([], pa.float64())}))
            continue
        nullable = route == "shape_changes_midstream" and i == 5
        cols = {"k": pa.array(rng.integers(0, groups, n).astype(np.int64) * mult - 17),
                "v": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0, mask=(rng.random(n) < 0.1) if nullable else None),
                "p": pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)}
        batches.append(pa.RecordBatch.from_pydict(cols))
    if route == "generic_program":
        funcs = [(O.MIN, "v", "lo"), (O.MAX, "v", "hi"), (O.COUNT_STAR, "", "n")]
    else:
        funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT, "v", "c"), (O.COUNT_STAR, "", "n")]
    predicate = {"on_input": ("v", ">", 64.0), "on_other": ("p", ">", 20.0), "none": None}[pred]
    names = batches[0].schema.names
    fspec = [(f, names.index(col) if col else None, pa.float64() if col else None) for f, col, _ in funcs]

    def run(stream_mode):
        agg = ops.DeviceAggregate(O.SINGLE, [pa.int64()], fspec, stream_mode=stream_mode)
        if predicate:
            agg.set_predicate(predicate[1], predicate[2])
        for i, b in enumerate(batches):
            kc, vc, pc = (DeviceColumn.from_arrow(b.column(j)) for j in range(3))
            agg.next([kc], [vc if col else None for _, col, _ in funcs], pred={"v": vc, "p": pc}[predicate[0]] if predicate else None,
                     nrows=b.num_rows)
            if stream_mode and i == 6:
                agg.sync()                                  # a sync point in the middle of the stream
        dcols = agg.result_device([0])
        res = agg.result_arrays([0], ["k"], [f[2] for f in funcs])
        dev = pa.RecordBatch.from_arrays([c.to_arrow() for c in dcols], names=res.schema.names)
        util.assert_batches_equal(dev, res, key_names=["k"], what="device finalisation vs host finalisation")
        agg.close()
        return res

    L.lib().vnm_set_profiling(1)
    got = run(True)
    p1, scan = _launches(b"agg_part_scatter1"), _launches(b"agg_scan")
    L.lib().vnm_set_profiling(0)
    # the segment routes: one launch per sync point (the sync after the seventh batch, the empty batch at the end) -- not one per batch
    if route in ("dense_two_level", "dense_one_level", "dense_split_final"):
        assert p1 == 2 and scan == 0, (p1, scan)
    elif route == "lds_scan_g7":
        assert p1 == 0 and scan == 2, (p1, scan)
    exp = _oracle(O.SINGLE, ["k"], funcs, batches, predicate)
    util.assert_agg_equal(got, exp, funcs, ["k"], what=f"async stream, {route}, pred {pred}")
    util.assert_agg_equal(run(False), exp, funcs, ["k"], what=f"the same stream batch by batch, {route}, pred {pred}")


MAN = util.manifest()


@pytest.mark.parametrize("case", MAN["sort_mixed"], ids=lambda c: c["name"])
def test_vinum_lib_sort_with_non_numeric_columns(case):
    """VERDICT r03 missing #1 (a17): Sort over record batches that carry string / binary / boolean / decimal columns -- as payload of
    `select * ... order by total` and as sort keys (`order by city_from desc, total asc`) -- through vinum_lib.Sort, against outputs
    of the reference's own Sort (sort.cpp:15-63) for the reference's orderby_queries shapes (test_query_results.py:627-745) and its
    NULL / NaN ordering cases (:1252-1266)."""
    from vinum_amd import vinum_lib as vl
    table = util.read_ipc(case["input"])
    expected = util.read_ipc(case["expected"])
    s = vl.Sort(case["cols"], [vl.SortOrder.DESC if o else vl.SortOrder.ASC for o in case["orders"]] if hasattr(vl, "SortOrder") else case["orders"])
    for b in util.sliced_batches(table, case["chunk"]):
        s.next(b)
    got = s.sorted()
    assert got.schema.equals(expected.schema), (got.schema, expected.schema)
    util.assert_batches_equal(got, expected, what=case["name"])   # every column, order-sensitive, bit-exact


@pytest.mark.parametrize("seed", range(8))
def test_vinum_lib_sort_mixed_columns_vs_oracle(seed):
    """Seeded: 1-3 sort keys drawn from string / large_string / binary / int / float / date columns (NULLs, NaN, ties), random
    directions, string + bool + decimal payload, optional LIMIT -- vinum_lib.Sort against the oracle (Arrow SortIndices + Take)."""
    import decimal
    from oracle import oracle as O
    from vinum_amd import vinum_lib as vl
    rng = np.random.default_rng(900 + seed)
    n = int(rng.choice([1, 7, 1000, 40_000, 250_000]))
    vocab = np.array([f"w{int(x):05d}"[: int(rng.integers(1, 7))] for x in rng.integers(0, 99999, 400)] + ["", "ä", "zz"], dtype=object)
    cols = {
        "rowid": pa.array(np.arange(n, dtype=np.int64)),
        "s": pa.array(vocab[rng.integers(0, len(vocab), n)], type=pa.string(), mask=rng.random(n) < 0.05),
        "ls": pa.array(vocab[rng.integers(0, 30, n)], type=pa.large_string()),
        "b": pa.array([bytes(x) for x in rng.integers(0, 4, (n, 2)).astype(np.uint8)], type=pa.binary(), mask=rng.random(n) < 0.05),
        "f": pa.array(np.where(rng.random(n) < 0.03, np.nan, np.round(rng.normal(0, 3, n), 1)), mask=rng.random(n) < 0.05),
        "i": pa.array(rng.integers(-3, 3, n).astype(np.int16), mask=rng.random(n) < 0.05),
        "flag": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1),
        "dec": pa.array([decimal.Decimal(int(x)).scaleb(-1) for x in rng.integers(-50, 50, n)], type=pa.decimal128(8, 1)),
    }
    t = pa.table(cols)
    keys = [str(k) for k in rng.choice(["s", "ls", "b", "f", "i", "dec"], size=int(rng.integers(1, 4)), replace=False)]
    orders = [int(rng.integers(0, 2)) for _ in keys]
    limit = int(rng.choice([0, 0, 5, max(1, n // 3)]))
    batches = util.sliced_batches(t, int(rng.choice([max(1, n // 3), 10_000, n + 1])))
    s = vl.Sort(keys, orders)
    o = O.OracleSort(keys, orders)
    for b in batches:
        s.next(b)
        o.next(b)
    got, exp = s.sorted(limit), o.sorted()
    if limit:
        exp = exp.slice(0, min(limit, n))
    util.assert_batches_equal(got, exp, what=f"seed {seed}: order by {keys} {orders} limit {limit}, {n} rows")


def test_vinum_lib_sort_boolean_key_is_rejected_like_the_reference():
    """vinum/core/algebra.py:191-201: sorting BY a boolean column is an error (Arrow 3.0 cannot); as payload it is fine."""
    from vinum_amd import vinum_lib as vl
    t = pa.table({"x": pa.array([3, 1, 2], pa.int64()), "flag": pa.array([True, None, False])})
    s = vl.Sort(["flag"], [0])
    s.next(t.to_batches()[0])
    with pytest.raises(RuntimeError):
        s.sorted()
    s = vl.Sort(["x"], [1])
    s.next(t.to_batches()[0])
    assert s.sorted().to_pydict() == {"x": [3, 2, 1], "flag": [True, False, None]}


@pytest.mark.parametrize("program", ["hot", "hot_no_pred", "minmax_count_star", "nullable_sum", "count_star"])
@pytest.mark.parametrize("groups", [10_000, 14_000, 25_000, 31_000])
@pytest.mark.parametrize("batches", [1, 3])
def test_dense_path_32_partitions_for_ranges_of_2e14_2e15_codes(program, groups, batches, monkeypatch):
    """Round 4 (VERDICT r03 #5a, the G ~ 1e4 cliff): ranges of 2^14 / 2^15 codes go through 32 ring-scatter partitions of 2^9 / 2^10
    slots (split final pass + merge) instead of 8 / 16 partitions through the tile-sorting scatter.  Bit-exact against the oracle for
    the hot program, generic programs, a nullable value column; one batch and a stream of three (deferred final pass)."""
    from oracle import oracle as O
    from vinum_amd import _lib as L
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(groups + len(program) + batches)
    n = 1_300_000
    k = rng.integers(0, groups, n).astype(np.int64) - groups // 3
    vmask = (rng.random(n) < 0.1) if program == "nullable_sum" else None
    t = pa.table({"k": pa.array(k), "v": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0, mask=vmask)})
    funcs = {"hot": [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT_STAR, "", "n")],
             "hot_no_pred": [(O.SUM, "v", "s"), (O.COUNT, "v", "c")],
             "minmax_count_star": [(O.MIN, "v", "lo"), (O.MAX, "v", "hi"), (O.COUNT_STAR, "", "n")],
             "nullable_sum": [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT_STAR, "", "n")],
             "count_star": [(O.COUNT_STAR, "", "n")]}[program]
    pred = None if program in ("hot_no_pred", "count_star") else ("v", ">", 64.0)
    bl = util.sliced_batches(t, (n // batches + 2) & ~1)
    L.lib().vnm_set_profiling(1)
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, bl, predicate=pred)
    p1, scan = _launches(b"agg_part_scatter1"), _launches(b"agg_scan")
    L.lib().vnm_set_profiling(0)
    assert p1 == len(bl), (p1, scan)             # the dense scatter took every batch
    util.assert_agg_equal(got, _oracle(O.SINGLE, ["k"], funcs, bl, pred), funcs, ["k"], what=f"G={groups} {program} x{batches}")


@pytest.mark.parametrize("ranks", [2, 3])
def test_string_group_keys_across_simulated_ranks(ranks):
    """f3 x e (VERDICT r03 missing #4): every simulated rank encodes its rows with its OWN device string dictionary
    (KeyDictionary / vnm_strdict_encode) and aggregates by code; union_of_dictionaries + rekey_codes turn the partial groups' key words
    into ids of one dictionary, the partial states merge into one operator (vnm_agg_merge_device) and decode through the union: equal
    to GenericHashAggregate over all rows on one rank (generic_hash_aggregate.h:10-45)."""
    import torch
    from vinum_amd import _lib as L, ops, distributed as D, vinum_lib as vl
    from vinum_amd.device import DeviceColumn
    rng = np.random.default_rng(31 + ranks)
    vocab = np.array([f"k{int(i):05d}" for i in range(20_000)] + ["", "ü", "zz"], dtype=object)
    n = 300_000
    tables = []
    for r in range(ranks):
        pick = rng.permutation(len(vocab))[: 12_000 + 2000 * r]
        tables.append(pa.table({"city": pa.array(vocab[pick[rng.integers(0, len(pick), n)]], type=pa.string(), mask=rng.random(n) < 0.02),
                                "v": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)}))
    fspec = [(L.SUM, 1, pa.float64()), (L.COUNT_STAR, None, None)]
    aggs, dicts = [], []
    for t in tables:
        kd = vl.KeyDictionary(pa.string())
        codes = kd.encode(t.column(0))
        a = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int32()], fspec)
        vc = DeviceColumn.from_arrow(t.column(1))
        a.next([DeviceColumn.from_arrow(codes)], [vc, None], nrows=n)
        aggs.append(a); dicts.append(kd)
    parts = [kd.values_by_code() for kd in dicts]
    merged = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int32()], fspec)
    union = None
    keep = []
    for r, a in enumerate(aggs):
        ng = a.finish()
        kp, ap_ = a.dense_ptrs()
        union, remap = D.union_of_dictionaries(parts, r)
        codes_w = torch.as_tensor(D._RawView(kp[0], ng), device="cuda")
        mask_w = torch.as_tensor(D._RawView(kp[1], ng), device="cuda")
        rek = D.rekey_codes(codes_w, mask_w, remap).contiguous()
        keep.append(rek)
        merged.merge(ng, [rek.data_ptr(), kp[1]], ap_)
    res = merged.result_arrays([0], ["city"], ["s", "n"])
    got = pa.RecordBatch.from_arrays([union.take(res.column(0)), res.column(1), res.column(2)], names=["city", "s", "n"])
    one = vl.GenericHashAggregate(["city"], ["city"], [vl.AggFuncDef(vl.SUM, "v", "s"), vl.AggFuncDef(vl.COUNT_STAR, "", "n")])
    for t in tables:
        one.next(t.to_batches()[0])
    exp = one.result()
    g = pa.Table.from_batches([got]).sort_by("city").combine_chunks()
    e = pa.Table.from_batches([exp]).sort_by("city").combine_chunks()
    assert g.num_rows == e.num_rows and g.column("city").equals(e.column("city"))          # the same groups (NULL among them)
    for c in ("s", "n"):
        util.assert_col_equal(g.column(c), e.column(c), f"{ranks} simulated ranks, string keys: {c}")   # bit-exact (quantised values)
    for a in aggs:
        a.close()
    merged.close()


@pytest.mark.parametrize("pred", ["on_input", "on_other", "none"])
@pytest.mark.parametrize("route", ["dense_two_level", "dense_one_level", "dense_32_partitions", "sparse_keys_fall_back", "all_keys_null_batch"])
@pytest.mark.parametrize("hint", [0, 1])
def test_nullable_key_through_the_dense_path(route, pred, hint, monkeypatch):
    """VERDICT r03 #5c: a NULLABLE single 8-byte key under the hot program goes through the dense path as it is -- pass 1 reads the
    key's validity (dring_scatter_kernel<KN>) and sums the NULL-key rows into the NULL slot of the operator's table; the fused result
    columns append that group last with a validity bitmap (single_numerical_hash_aggregate.cpp:24-32).  Garbage under the NULL slots of
    the key column, the real key 0 next to the NULL group, sparse keys (the dense path declines -> the packed route), a batch whose
    keys are ALL NULL, several batches."""
    from oracle import oracle as O
    from vinum_amd import _lib as L
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    groups = {"dense_two_level": 1_500_000, "dense_one_level": 900_000, "dense_32_partitions": 20_000, "sparse_keys_fall_back": 300_000,
              "all_keys_null_batch": 900_000}[route]
    if route == "dense_two_level":
        monkeypatch.setenv("VNM_DENSE_ONE_LEVEL", "0")
    rng = np.random.default_rng(len(route) * 3 + len(pred) + hint)
    batches = []
    for bi, n in enumerate([1_100_000, 400_001, 650_000]):
        k = rng.integers(0, groups, n).astype(np.int64) * (1_000_003 if route == "sparse_keys_fall_back" else 1)
        k[::997] = 0
        mask = rng.random(n) < 0.12
        if route == "all_keys_null_batch" and bi == 1:
            mask[:] = True
        k[mask] = rng.integers(-2**40, 2**40, int(mask.sum()))        # garbage under the NULL slots: it must never be looked at
        batches.append(pa.RecordBatch.from_pydict({
            "k": pa.array(k, mask=mask),
            "v": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0),
            "p": pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)}))
    funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT, "v", "c"), (O.COUNT_STAR, "", "n")]
    predicate = {"on_input": ("v", ">", 64.0), "on_other": ("p", ">", 20.0), "none": None}[pred]
    L.lib().vnm_set_profiling(1)
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, batches, predicate=predicate, expected_groups=groups if hint else 0)
    pack, p1 = _launches(b"agg_pack_keys"), _launches(b"agg_part_scatter1")
    L.lib().vnm_set_profiling(0)
    if route.startswith("dense"):
        assert pack == 0 and p1 == len(batches), (pack, p1)      # no packing pass: the dense scatter read the nullable key itself
    exp = _oracle(O.SINGLE, ["k"], funcs, batches, predicate)
    util.assert_agg_equal(got, exp, funcs, ["k"], what=f"nullable key, {route}, pred {pred}, hint {hint}")
    if route.startswith("dense"):
        assert got.column(0)[got.num_rows - 1].as_py() is None    # the NULL group comes last, as in the reference's single-key operator


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VNM_FUZZ_SEEDS", "64")))))
def test_random_round4_paths_vs_oracle(seed, monkeypatch):
    """Seeded differential test over what round 4 added to the hot shape, in combination: nullable keys (garbage under the NULLs) through
    the dense path's pass 1, stream mode (segments of one launch) against synchronous calls, skewed keys whose spills are folded into
    the fused result columns, the 32-partition geometry of small ranges, short batches joining a pending pass, empty and ragged
    batches.  Result columns finalised on the device and the host finaliser over the partial state must both equal the oracle."""
    from oracle import oracle as O
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    rng = np.random.default_rng(77_000 + seed)
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", str(int(rng.choice([50_000, 100_000]))))
    groups = int(rng.choice([7, 900, 6_000, 12_000, 28_000, 60_000, 400_000, 1_500_000]))
    if rng.random() < 0.3:
        monkeypatch.setenv("VNM_DENSE_ONE_LEVEL", "0")
    nb = int(rng.choice([1, 2, 3, 5]))
    sizes = [int(rng.choice([0, 77, 8192, 120_001, 300_000, 700_000])) for _ in range(nb)]
    sizes[0] = int(rng.choice([250_000, 600_000, 900_001]))
    nullable_key = rng.random() < 0.45
    skew = rng.random() < 0.35
    lo = int(rng.choice([0, -groups // 2, 10**11]))
    stream_mode = bool(rng.random() < 0.5)
    batches = []
    for n in sizes:
        u = rng.random(n)
        k = (np.floor((u ** 4 if skew else u) * groups)).astype(np.int64) + lo
        mask = None
        if nullable_key and n:
            mask = rng.random(n) < rng.choice([0.02, 0.3])
            k[mask] = rng.integers(-2**50, 2**50, int(mask.sum()))
        batches.append(pa.RecordBatch.from_pydict({
            "k": pa.array(k, mask=mask) if n else pa.array([], pa.int64()),
            "v": pa.array(rng.integers(-2**13, 2**13, n).astype(np.float64) / 128.0),
            "p": pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)}))
    funcs = [(O.SUM, "v", "s"), (O.AVG, "v", "a"), (O.COUNT, "v", "c"), (O.COUNT_STAR, "", "n")]
    funcs = [funcs[i] for i in sorted(rng.choice(4, size=int(rng.integers(1, 5)), replace=False))]
    if not any(f[1] for f in funcs):
        funcs.append((O.SUM, "v", "s"))
    predicate = [("v", ">", 3.0), ("p", ">", 20.0), None][int(rng.integers(0, 3))]
    hint = int(rng.choice([0, 0, groups]))
    names = ["k", "v", "p"]
    fspec = [(f, names.index(col) if col else None, pa.float64() if col else None) for f, col, _ in funcs]
    agg = ops.DeviceAggregate(O.SINGLE, [pa.int64()], fspec, expected_groups=hint, stream_mode=stream_mode)
    if predicate:
        agg.set_predicate(predicate[1], predicate[2])
    for b in batches:
        kc, vc, pc = (DeviceColumn.from_arrow(b.column(j)) for j in range(3))
        agg.next([kc], [vc if col else None for _, col, _ in funcs], pred={"v": vc, "p": pc}[predicate[0]] if predicate else None, nrows=b.num_rows)
    what = f"seed {seed}: G~{groups} lo={lo} nullable_key={nullable_key} skew={skew} stream={stream_mode} hint={hint} pred={predicate} sizes={sizes}"
    dcols = agg.result_device([0])
    res = agg.result_arrays([0], ["k"], [f[2] for f in funcs])
    dev = pa.RecordBatch.from_arrays([c.to_arrow() for c in dcols], names=res.schema.names)
    util.assert_batches_equal(dev, res, key_names=["k"], what=what + ": device result columns vs host finalisation")
    agg.close()
    util.assert_agg_equal(res, _oracle(O.SINGLE, ["k"], funcs, batches, predicate), funcs, ["k"], what=what)


@pytest.mark.parametrize("program", ["sum_sum", "sum_avg_counts", "avg_only_second"])
@pytest.mark.parametrize("pred", ["on_first", "on_other", "none"])
@pytest.mark.parametrize("groups,levels", [(1_500_000, "default"), (3_000_000, "default"), (1_200_000, "sample_misses"), (1_400_000, "two_batches")])
def test_dense_path_two_input_columns(program, pred, groups, levels, monkeypatch):
    """`SELECT k, sum(a), sum(b) ...` over two plain float64 columns on the dense path -- two-value entries through the ring
    scatter (dring_scatter_kernel<V2>) and a final pass with two compensated sums per slot (dpart_final2_kernel) -- bit-exact
    against the oracle; a key the range sample never saw fails the pass (no spill buffer for two-value entries) and the batch
    takes the hash partitions: same result.  agg_func_factory.cpp:108-176 (one accumulator per function and column)."""
    from oracle import oracle as O
    from vinum_amd import _lib as L
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(groups % 1000 + len(program) + len(pred))
    n = 1_400_000
    k = rng.integers(0, groups, n).astype(np.int64) - groups // 5
    if levels == "sample_misses":
        k[7::200_003] = 90_000_000
    t = pa.table({"k": pa.array(k), "a": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0),
                  "b": pa.array(rng.integers(-2**13, 2**13, n).astype(np.float64) / 64.0),
                  "p": pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)})
    funcs = {"sum_sum": [(O.SUM, "a", "sa"), (O.SUM, "b", "sb")],
             "sum_avg_counts": [(O.SUM, "a", "sa"), (O.AVG, "b", "ab"), (O.COUNT, "a", "ca"), (O.COUNT, "b", "cb"), (O.COUNT_STAR, "", "n")],
             "avg_only_second": [(O.COUNT, "a", "ca"), (O.AVG, "b", "ab")]}[program]
    predicate = {"on_first": ("a", ">", 64.0), "on_other": ("p", ">", 20.0), "none": None}[pred]
    bl = util.sliced_batches(t, n if levels != "two_batches" else 800_000)
    L.lib().vnm_set_profiling(1)
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, bl, predicate=predicate)
    p1, p2 = _launches(b"agg_part_scatter1"), _launches(b"agg_part_scatter2")
    L.lib().vnm_set_profiling(0)
    if levels == "default":
        assert p1 == 1 and p2 == 1, (p1, p2)      # the two-value dense path took the batch (two scatter levels, one launch each)
    util.assert_agg_equal(got, _oracle(O.SINGLE, ["k"], funcs, bl, predicate), funcs, ["k"], exact_float_inputs=("a", "b"),
                          what=f"two columns, {program}, pred {pred}, G={groups} {levels}")


@pytest.mark.parametrize("program", ["sum_sum", "three_columns_mixed", "minmax_nullable", "int_columns"])
@pytest.mark.parametrize("pred", ["on_first", "on_other", "none"])
@pytest.mark.parametrize("groups,shape", [(1000, "one_batch"), (5000, "one_batch"), (3000, "three_batches"), (4000, "null_keys_later"),
                                          (3000, "hinted"), (5000, "sparse_keys")])
def test_small_range_many_columns_split_per_column(program, pred, groups, shape, monkeypatch):
    """Round 4: a few thousand groups in a small key range under two or more 8-byte input columns -- too many groups for the hashed
    LDS table of the multi-column scan -- are aggregated one input column at a time by the direct-addressed LDS scan (make_parts with one
    column per part, joined by key at the end) instead of through wide partition entries.  Bit-exact against the oracle;
    a NULL key in a later batch, several batches, a caller's hint; sparse keys of the same count keep the old path.
    agg_func_factory.cpp:108-176 (every function has its own accumulator: the order of the columns' passes cannot matter)."""
    from oracle import oracle as O
    from vinum_amd import _lib as L
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(groups + len(program) * 7 + len(pred) + len(shape))
    n = 900_000
    k = rng.integers(0, groups, n).astype(np.int64) - groups // 4
    if shape == "sparse_keys":
        k = k * 1_000_003
    kmask = None
    if shape == "null_keys_later":
        kmask = np.zeros(n, dtype=bool)
        kmask[n // 2:] = rng.random(n - n // 2) < 0.05
    cmask = (rng.random(n) < 0.1) if program == "minmax_nullable" else None
    t = pa.table({"k": pa.array(k, mask=kmask),
                  "a": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0),
                  "b": pa.array(rng.integers(-2**13, 2**13, n).astype(np.float64) / 64.0, mask=cmask),
                  "c": pa.array(rng.integers(-1000, 1000, n).astype(np.int64)),
                  "d": pa.array(rng.integers(0, 1 << 40, n).astype(np.uint64)),
                  "p": pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)})
    funcs = {"sum_sum": [(O.SUM, "a", "sa"), (O.SUM, "b", "sb")],
             "three_columns_mixed": [(O.SUM, "a", "sa"), (O.AVG, "b", "ab"), (O.COUNT_STAR, "", "n"), (O.MAX, "c", "mc"), (O.COUNT, "a", "ca")],
             "minmax_nullable": [(O.MIN, "a", "la"), (O.MAX, "b", "hb"), (O.SUM, "b", "sb"), (O.COUNT, "b", "cb")],
             "int_columns": [(O.SUM, "c", "sc"), (O.MIN, "d", "ld"), (O.AVG, "c", "ac"), (O.COUNT_STAR, "", "n")]}[program]
    predicate = {"on_first": ("a", ">", 64.0), "on_other": ("p", ">", 20.0), "none": None}[pred]
    bl = util.sliced_batches(t, 300_000 if shape == "three_batches" else (450_000 if shape == "null_keys_later" else n))
    L.lib().vnm_set_profiling(1)
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, bl, predicate=predicate, expected_groups=groups if shape == "hinted" else 0)
    joins, p1 = _launches(b"agg_split_join"), _launches(b"agg_part_scatter1")
    L.lib().vnm_set_profiling(0)
    if shape == "sparse_keys":
        assert joins == 0, joins
    elif groups >= 3000:
        assert joins == 1, (joins, p1)   # one part per column (1000 groups still fit the hashed table of a two-column program; a part
                                         # with a generic program may take the four-partition form of the small range)
    util.assert_agg_equal(got, _oracle(O.SINGLE, ["k"], funcs, bl, predicate), funcs, ["k"], exact_float_inputs=("a", "b"),
                          what=f"split per column: {program}, pred {pred}, G={groups} {shape}")


_SPLIT_SHAPES = [(300_000, "one_batch"), (1_500_000, "one_batch"), (1_500_000, "pairs"), (1_200_000, "two_batches"), (1_300_000, "null_keys_later"),
                 (1_400_000, "heavy_keys"), (1_500_000, "sparse_keys"), (1_100_000, "two_batches_bounded")]
_SPLIT_CASES = ([("three_sums", "on_first") + sh for sh in _SPLIT_SHAPES] + [("four_columns_mixed", "on_first") + sh for sh in _SPLIT_SHAPES] +
                [("nullable_and_minmax", "none") + sh for sh in _SPLIT_SHAPES[:6]] + [("three_sums", "none") + sh for sh in _SPLIT_SHAPES[1:4]] +
                [("two_sums", "on_first") + sh for sh in _SPLIT_SHAPES[:4]] + [("two_minmax", "none") + sh for sh in (_SPLIT_SHAPES[0], _SPLIT_SHAPES[3], _SPLIT_SHAPES[4])])


@pytest.mark.parametrize("program,pred,groups,shape", _SPLIT_CASES)
def test_many_groups_many_columns_split_over_the_dense_path(program, pred, groups, shape, monkeypatch):
    """Round 4: three or more 8-byte input columns over a key the dense path takes are aggregated per column (or per pair of float64
    columns under sums / counts: two-value entries) by the dense path and joined at the end -- by a plain copy when every part
    wrote its groups in the same order (same code map), by the sort join otherwise (heavy keys that spilled to the side table,
    a NULL-key group from the general scan).  Bit-exact against the oracle.  agg_func_factory.cpp:108-176."""
    from oracle import oracle as O
    from vinum_amd import _lib as L
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    monkeypatch.setenv("VNM_AGG_SPLIT_DENSE_MIN_ROWS", "100000")
    if shape == "pairs":
        monkeypatch.setenv("VNM_AGG_SPLIT_PAIRS_MIN_GROUPS", "1000000")
        monkeypatch.setenv("VNM_AGG_PAIRS_MIN_ROWS", "100000")     # (pairs are for batches of 2^27 rows and more)
    if shape == "two_batches_bounded":
        monkeypatch.setenv("VNM_SPLIT_PENDING_BYTES", "1")    # (the parts run their final passes per batch instead of keeping the scatter output)
    rng = np.random.default_rng(groups % 977 + len(program) * 5 + len(pred) + len(shape))
    n = 2_400_000 if shape.startswith("two_batches") else 1_500_000    # (a batch joins a waiting final pass from ~1e6 rows on)
    k = rng.integers(0, groups, n).astype(np.int64) - groups // 4
    if shape == "heavy_keys":
        k[rng.random(n) < 0.3] = 17   # one key holds ~30 % of the rows
    if shape == "sparse_keys":
        k = k * 1_000_003
    kmask = None
    if shape == "null_keys_later":
        kmask = np.zeros(n, dtype=bool)
        kmask[n // 2:] = rng.random(n - n // 2) < 0.05
    cmask = (rng.random(n) < 0.1) if program == "nullable_and_minmax" else None
    t = pa.table({"k": pa.array(k, mask=kmask),
                  "a": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0),
                  "b": pa.array(rng.integers(-2**13, 2**13, n).astype(np.float64) / 64.0, mask=cmask),
                  "c": pa.array(rng.integers(0, 2**10, n).astype(np.float64) / 8.0),
                  "d": pa.array(rng.integers(-1000, 1000, n).astype(np.int64))})
    funcs = {"three_sums": [(O.SUM, "a", "sa"), (O.SUM, "b", "sb"), (O.AVG, "c", "ac"), (O.COUNT_STAR, "", "n")],
             "four_columns_mixed": [(O.SUM, "a", "sa"), (O.AVG, "b", "ab"), (O.MAX, "d", "md"), (O.COUNT, "c", "cc"), (O.MIN, "c", "lc"), (O.COUNT_STAR, "", "n")],
             "nullable_and_minmax": [(O.MIN, "a", "la"), (O.SUM, "b", "sb"), (O.COUNT, "b", "cb"), (O.SUM, "d", "sd")],
             # TWO columns: one part per column below ~1.5e6 groups (the two-value entries take over above), always for programs they do not carry
             "two_sums": [(O.SUM, "a", "sa"), (O.AVG, "c", "ac"), (O.COUNT_STAR, "", "n")],
             "two_minmax": [(O.MIN, "a", "la"), (O.MAX, "d", "hd"), (O.COUNT, "a", "ca")]}[program]
    predicate = {"on_first": ("a", ">", 64.0), "none": None}[pred]
    bl = util.sliced_batches(t, 1_200_000 if shape.startswith("two_batches") else (750_000 if shape == "null_keys_later" else n))
    L.lib().vnm_set_profiling(1)
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, bl, predicate=predicate)
    joins, sorts = _launches(b"agg_split_join"), _launches(b"agg_split_sort")
    routes = {nm.decode(): _launches(nm) for nm in (b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final", b"agg_split_units")}
    L.lib().vnm_set_profiling(0)
    if shape == "sparse_keys" or (program == "two_sums" and shape == "pairs"):   # (two sum columns from the pairs' threshold on: the two-value entries directly)
        assert joins == 0, joins
    elif program == "two_sums" and groups >= 1_400_000:
        assert joins in (0, 1), joins    # (around the 1.5e6-group threshold the estimate decides)
    else:
        assert joins == 1, joins
    if (shape in ("one_batch", "pairs") and joins) or (shape == "two_batches" and program in ("three_sums", "two_sums")):
        assert sorts == 0, (sorts, routes)   # joined by units of 64 codes: no sort, no gather (generic programs merge their batches' runs
                                        # through the table: another order, the sort join)
    util.assert_agg_equal(got, _oracle(O.SINGLE, ["k"], funcs, bl, predicate), funcs, ["k"], exact_float_inputs=("a", "b", "c"),
                          what=f"dense split: {program}, pred {pred}, G={groups} {shape}")


@pytest.mark.parametrize("ncols", [3, 4, 5, 6])
@pytest.mark.parametrize("pred", ["on_input", "on_other", "none"])
@pytest.mark.parametrize("groups,shape", [(7, "one_batch"), (300, "three_batches"), (5, "ragged_tail"), (2000, "saturates_lds"), (40, "empty_sentinel_key")])
def test_scan_over_three_to_six_float_columns(ncols, pred, groups, shape, monkeypatch):
    """Round 4: `SELECT k, sum(a), avg(b), count(c), ..., count(*) GROUP BY k` over few groups and three to six plain float64 columns
    takes agg_hotn_kernel (16-byte loads, next tile prefetched, ONE count atomic per row: the other count words are copied from it
    before every flush) instead of the interpreted scan.  Bit-exact against the oracle: one and several batches, a ragged tail
    (scalar path), more groups than the LDS table takes (saturated keys go to the HBM table row by row), the key that equals the
    table's EMPTY sentinel.  agg_func_factory.cpp:108-176; base_aggregate.cpp:23-45."""
    from oracle import oracle as O
    from vinum_amd import _lib as L
    rng = np.random.default_rng(ncols * 31 + groups + len(pred) + len(shape))
    n = {"one_batch": 400_000, "three_batches": 390_000, "ragged_tail": 100_003, "saturates_lds": 300_000, "empty_sentinel_key": 65_538}[shape]
    k = rng.integers(0, groups, n).astype(np.int64) * 977 - 1234
    if shape == "empty_sentinel_key":
        k[::7] = -1        # (0xFFFF...: the LDS / HBM tables' EMPTY tag has a slot of its own)
    cols = {"k": pa.array(k)}
    names = "abcdef"[:ncols]
    for i, c in enumerate(names):
        cols[c] = pa.array(rng.integers(-2**13, 2**13, n).astype(np.float64) / (64.0 * 2**i))   # (dyadic: sums are exact in any order)
    cols["p"] = pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)
    t = pa.table(cols)
    kinds = [O.SUM, O.AVG, O.COUNT, O.SUM, O.AVG, O.SUM]
    funcs = [(kinds[i], c, f"f_{c}") for i, c in enumerate(names)] + [(O.COUNT_STAR, "", "n"), (O.SUM, "b", "sb2"), (O.COUNT, "a", "ca")]
    predicate = {"on_input": ("b", ">", 0.0), "on_other": ("p", ">", 20.0), "none": None}[pred]
    bl = util.sliced_batches(t, 130_000 if shape == "three_batches" else n)
    L.lib().vnm_set_profiling(1)
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, bl, predicate=predicate)
    scans = _launches(b"agg_scan")
    L.lib().vnm_set_profiling(0)
    assert scans >= len(bl), scans
    util.assert_agg_equal(got, _oracle(O.SINGLE, ["k"], funcs, bl, predicate), funcs, ["k"], exact_float_inputs=tuple(names),
                          what=f"hotn: {ncols} columns, pred {pred}, G={groups} {shape}")


@pytest.mark.parametrize("pred", ["on_input", "none"])
@pytest.mark.parametrize("route", ["dense_per_column", "dense_range_too_wide_for_one_batch", "small_range_per_column", "few_groups_scan", "sparse_keys",
                                   "mixed_types", "two_columns", "shape_changes_midstream"])
def test_async_stream_with_several_input_columns(route, pred, monkeypatch):
    """Stream mode over SEVERAL input columns (round 4): the operator records the batches and chooses the path from the stream's total
    row count -- a code range too wide for one 2^24-row batch is fine for the stream (G = 1e8, three columns, 30 batches: 315 ->
    31 ms) -- cuts the program into parts when a rule applies and hands them the batches one by one; the parts record them in turn
    (one launch per part and sync point) or, for programs the segment kernels do not take, run them right away.  Same result as
    the oracle fed batch by batch and as the synchronous mode; ragged / tiny / empty batches, a second sync point, a batch of another
    shape in the middle.  base_aggregate.cpp:23-45."""
    from oracle import oracle as O
    from vinum_amd import _lib as L, ops
    from vinum_amd.device import DeviceColumn
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    monkeypatch.setenv("VNM_AGG_SPLIT_DENSE_MIN_ROWS", "1000000")
    # round 6: two / three fixed-point columns go through ONE pass (vnm_agg_fxn.inc) before the program is cut per column; the per-column
    # routes this test was written for are kept under test with that path switched off (pred == "none"), the one-pass path with it on
    one_pass = pred == "on_input" and route in ("dense_per_column", "dense_range_too_wide_for_one_batch", "two_columns")
    if not one_pass:
        monkeypatch.setenv("VNM_DENSE_FXN", "0")
    groups = {"dense_per_column": 300_000, "dense_range_too_wide_for_one_batch": 1_200_000, "small_range_per_column": 3_000, "few_groups_scan": 7,
              "sparse_keys": 200_000, "mixed_types": 400_000, "two_columns": 500_000, "shape_changes_midstream": 300_000}[route]
    mult = 7919 if route == "sparse_keys" else 1
    rng = np.random.default_rng(len(route) * 17 + len(pred))
    sizes = [600_000, 262_144, 300_001, 77, 8192, 123_457, 499_999, 16_384 + 2, 650_000, 0]
    if route == "dense_range_too_wide_for_one_batch":    # 2^21 codes: 13 times a batch's rows, half the stream's (one sync point: the result)
        sizes = [160_000] * 23 + [160_001]
    batches = []
    for i, n in enumerate(sizes):
        nullable = route == "shape_changes_midstream" and i == 5
        cols = {"k": pa.array(rng.integers(0, groups, n).astype(np.int64) * mult - 17),
                "a": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0),
                "b": pa.array(rng.integers(-2**13, 2**13, n).astype(np.float64) / 64.0, mask=(rng.random(n) < 0.1) if nullable else None),
                "c": pa.array(rng.integers(0, 2**10, n).astype(np.float64) / 8.0),
                "d": pa.array(rng.integers(-1000, 1000, n).astype(np.int64))}
        batches.append(pa.RecordBatch.from_pydict(cols))
    funcs = {"mixed_types": [(O.SUM, "a", "sa"), (O.MAX, "d", "md"), (O.AVG, "c", "ac"), (O.SUM, "d", "sd"), (O.COUNT_STAR, "", "n")],
             "two_columns": [(O.SUM, "a", "sa"), (O.AVG, "c", "ac"), (O.COUNT_STAR, "", "n")]}.get(
                 route, [(O.SUM, "a", "sa"), (O.AVG, "b", "ab"), (O.SUM, "c", "sc"), (O.COUNT, "b", "cb"), (O.COUNT_STAR, "", "n")])
    predicate = {"on_input": ("a", ">", 64.0), "none": None}[pred]
    names = batches[0].schema.names
    fspec = [(f, names.index(col) if col else None, batches[0].schema.field(col).type if col else None) for f, col, _ in funcs]

    def run(stream_mode):
        agg = ops.DeviceAggregate(O.SINGLE, [pa.int64()], fspec, stream_mode=stream_mode)
        if predicate:
            agg.set_predicate(predicate[1], predicate[2])
        for i, b in enumerate(batches):
            dc = {nm: DeviceColumn.from_arrow(b.column(j)) for j, nm in enumerate(names)}
            agg.next([dc["k"]], [dc[col] if col else None for _, col, _ in funcs], pred=dc[predicate[0]] if predicate else None, nrows=b.num_rows)
            if stream_mode and i == 6 and route != "dense_range_too_wide_for_one_batch":
                agg.sync()
        dcols = agg.result_device([0])
        res = agg.result_arrays([0], ["k"], [f[2] for f in funcs])
        dev = pa.RecordBatch.from_arrays([c.to_arrow() for c in dcols], names=res.schema.names)
        util.assert_batches_equal(dev, res, key_names=["k"], what="device finalisation vs host finalisation")
        agg.close()
        return res

    L.lib().vnm_set_profiling(1)
    got = run(True)
    p1, joins = _launches(b"agg_part_scatter1"), _launches(b"agg_split_join")
    routes = {nm.decode(): _launches(nm) for nm in (b"agg_scan", b"agg_part_scatter2", b"agg_part_final", b"agg_estimate")}
    L.lib().vnm_set_profiling(0)
    if one_pass:
        assert joins == 0 and p1 >= 1, (joins, p1, routes)   # no parts, no join (the short tail after the sync point takes a path of its own)
    elif route == "dense_per_column":
        assert joins == 1 and p1 == 2 * 3, (joins, p1, routes)      # three parts, one launch each per sync point -- not one per batch
    elif route == "dense_range_too_wide_for_one_batch":
        assert joins == 1 and p1 == 3, (joins, p1, routes)
    elif route == "two_columns":
        assert joins == 1 and p1 == 2 * 2, (joins, p1)
    elif route in ("few_groups_scan", "sparse_keys"):
        assert joins == 0, joins
    exp = _oracle(O.SINGLE, ["k"], funcs, batches, predicate)
    util.assert_agg_equal(got, exp, funcs, ["k"], exact_float_inputs=("a", "b", "c"), what=f"async stream, several columns, {route}, pred {pred}")
    util.assert_agg_equal(run(False), exp, funcs, ["k"], exact_float_inputs=("a", "b", "c"), what=f"the same stream batch by batch, {route}, pred {pred}")


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VNM_FUZZ_SEEDS_COLS", "48")))))
def test_random_programs_over_several_columns_vs_oracle(seed, monkeypatch):
    """Seeded differential test over the multi-column routes of round 4, in combination: 2 ... 7 input columns of float64 / int64 / uint64,
    some nullable, any function per column; few groups (agg_hotn_kernel or the interpreted scan), a few thousand in a small range (one LDS
    scan per column), many over a dense key (dense path per column or pair, unit join / sort join), sparse keys (wide entries); one to
    four ragged batches, synchronous or stream mode, heavy keys, NULL keys in a later batch, a predicate on an input or another column.
    Device-finalised result columns and the host finaliser over the partial state must both equal the oracle."""
    from oracle import oracle as O
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    rng = np.random.default_rng(91_000 + seed)
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    monkeypatch.setenv("VNM_AGG_SPLIT_DENSE_MIN_ROWS", str(int(rng.choice([200_000, 600_000]))))
    if rng.random() < 0.4:
        monkeypatch.setenv("VNM_AGG_SPLIT_PAIRS_MIN_GROUPS", "1000000")
        monkeypatch.setenv("VNM_AGG_PAIRS_MIN_ROWS", "200000")
    if rng.random() < 0.15:
        monkeypatch.setenv("VNM_SPLIT_PENDING_BYTES", "1")
    groups = int(rng.choice([5, 11, 300, 3_000, 7_000, 40_000, 400_000, 1_500_000]))
    ncols = int(rng.integers(2, 8))
    nb = int(rng.choice([1, 1, 2, 4]))
    sizes = [int(rng.choice([0, 77, 8192, 150_001, 400_000])) for _ in range(nb)]
    sizes[0] = int(rng.choice([300_000, 700_000, 1_000_001]))
    sparse = rng.random() < 0.2
    heavy = rng.random() < 0.2
    null_keys_later = nb > 1 and rng.random() < 0.25
    all_float = rng.random() < 0.5
    stream_mode = bool(rng.random() < 0.5)
    types = [pa.float64() if all_float or rng.random() < 0.6 else (pa.int64() if rng.random() < 0.7 else pa.uint64()) for _ in range(ncols)]
    nullable = [(not all_float) and rng.random() < 0.2 for _ in range(ncols)]
    names = ["k"] + [f"c{i}" for i in range(ncols)] + ["p"]
    batches = []
    for bi, n in enumerate(sizes):
        k = rng.integers(0, groups, n).astype(np.int64) * (1_000_003 if sparse else 1) - 5
        if heavy and n:
            k[rng.random(n) < 0.3] = 1
        kmask = (rng.random(n) < 0.05) if (null_keys_later and bi > 0 and n) else None
        cols = {"k": pa.array(k, mask=kmask)}
        for i in range(ncols):
            m = (rng.random(n) < 0.1) if nullable[i] and n else None
            if types[i] == pa.float64():
                cols[f"c{i}"] = pa.array(rng.integers(-2**13, 2**13, n).astype(np.float64) / (64.0 * 2 ** (i % 3)), mask=m)
            elif types[i] == pa.int64():
                cols[f"c{i}"] = pa.array(rng.integers(-2**40, 2**40, n).astype(np.int64), mask=m)
            else:
                cols[f"c{i}"] = pa.array(rng.integers(0, 2**41, n).astype(np.uint64), mask=m)
        cols["p"] = pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)
        batches.append(pa.RecordBatch.from_pydict(cols))
    kinds = [O.SUM, O.AVG, O.COUNT] if all_float else [O.SUM, O.AVG, O.COUNT, O.MIN, O.MAX]
    funcs = []
    for i in range(ncols):
        for f in rng.choice(kinds, size=int(rng.integers(1, 3)), replace=False):
            funcs.append((int(f), f"c{i}", f"f{len(funcs)}"))
    if rng.random() < 0.6:
        funcs.append((O.COUNT_STAR, "", "n"))
    funcs = funcs[:14]
    predicate = None
    r = rng.random()
    if r < 0.3:
        predicate = ("p", ">", 20.0)
    elif r < 0.6 and types[0] == pa.float64() and not nullable[0]:
        predicate = ("c0", ">", 0.0)
    hint = int(rng.choice([0, 0, groups]))
    fspec = [(f, names.index(col) if col else None, batches[0].schema.field(col).type if col else None) for f, col, _ in funcs]
    agg = ops.DeviceAggregate(O.SINGLE, [pa.int64()], fspec, expected_groups=hint, stream_mode=stream_mode)
    if predicate:
        agg.set_predicate(predicate[1], predicate[2])
    for b in batches:
        dc = {nm: DeviceColumn.from_arrow(b.column(j)) for j, nm in enumerate(names)}
        agg.next([dc["k"]], [dc[col] if col else None for _, col, _ in funcs], pred=dc[predicate[0]] if predicate else None, nrows=b.num_rows)
    what = (f"seed {seed}: G~{groups} C={ncols} types={[str(t) for t in types]} nullable={nullable} sparse={sparse} heavy={heavy} null_keys_later={null_keys_later} "
            f"stream={stream_mode} hint={hint} pred={predicate} sizes={sizes} funcs={[(f, c) for f, c, _ in funcs]}")
    dcols = agg.result_device([0])
    res = agg.result_arrays([0], ["k"], [f[2] for f in funcs])
    dev = pa.RecordBatch.from_arrays([c.to_arrow() for c in dcols], names=res.schema.names)
    util.assert_batches_equal(dev, res, key_names=["k"], what=what + ": device result columns vs host finalisation")
    agg.close()
    util.assert_agg_equal(res, _oracle(O.SINGLE, ["k"], funcs, batches, predicate), funcs, ["k"], exact_float_inputs=tuple(f"c{i}" for i in range(ncols)), what=what)


@pytest.mark.parametrize("ncols", [3, 5, 6])
@pytest.mark.parametrize("nulls", ["no_nulls", "nullable_columns", "nullable_predicate_input"])
@pytest.mark.parametrize("groups,shape", [(7, "one_batch"), (200, "two_batches"), (9, "ragged_tail"), (1500, "saturates_lds")])
def test_scan_over_several_columns_of_mixed_types(ncols, nulls, groups, shape):
    """agg_hotn_kernel also takes int64 / uint64 columns (SUM / AVG: the 128-bit sum's 32-bit lanes, agg_funcs.h:319-435) and MIN / MAX
    of any of the three types next to the float64 sums -- bit-exact against the oracle (few groups, several batches, the scalar tail,
    keys the LDS table cannot take).  Columns with validity bitmaps keep the interpreted scan (tried in agg_hotn_kernel: per-column
    validity bytes and own COUNT words made it slower than that scan, 5.5 against 4.6 ms per 5e8 rows) -- same shapes, same oracle."""
    from oracle import oracle as O
    rng = np.random.default_rng(ncols * 37 + groups + len(shape) + len(nulls))
    n = {"one_batch": 300_000, "two_batches": 260_000, "ragged_tail": 90_001, "saturates_lds": 200_000}[shape]
    cols = {"k": pa.array(rng.integers(0, groups, n).astype(np.int64) * 31 - 7)}
    funcs = []
    for i in range(ncols):
        t = i % 3
        m = (rng.random(n) < 0.15) if (nulls != "no_nulls" and i % 2 == 0) else None
        if t == 0:
            cols[f"c{i}"] = pa.array(rng.integers(-2**13, 2**13, n).astype(np.float64) / 64.0, mask=m)
            funcs += [(O.SUM, f"c{i}", f"s{i}"), (O.MAX, f"c{i}", f"hi{i}"), (O.COUNT, f"c{i}", f"cn{i}")]
        elif t == 1:
            cols[f"c{i}"] = pa.array(rng.integers(-2**62, 2**62, n).astype(np.int64), mask=m)       # (sums beyond 64 bits: decimal128 results)
            funcs += [(O.SUM, f"c{i}", f"s{i}"), (O.AVG, f"c{i}", f"a{i}"), (O.MIN, f"c{i}", f"lo{i}")]
        else:
            cols[f"c{i}"] = pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * 2 + 1, mask=m)
            funcs += [(O.MAX, f"c{i}", f"hi{i}"), (O.COUNT, f"c{i}", f"n{i}"), (O.SUM, f"c{i}", f"s{i}")]
    funcs.append((O.COUNT_STAR, "", "n"))
    cols["p"] = pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)
    t = pa.table(cols)
    bl = util.sliced_batches(t, 130_000 if shape == "two_batches" else n)
    pred = ("c0", ">", -50.0) if nulls == "nullable_predicate_input" else ("p", ">", 20.0)
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, bl, predicate=pred)
    util.assert_agg_equal(got, _oracle(O.SINGLE, ["k"], funcs, bl, pred), funcs, ["k"], exact_float_inputs=tuple(f"c{i}" for i in range(ncols)),
                          what=f"hotn mixed types: {ncols} columns, {nulls}, G={groups} {shape}")


@pytest.mark.parametrize("ncols", [7, 8, 10, 13])
@pytest.mark.parametrize("groups,shape", [(7, "one_batch"), (60, "two_batches"), (200, "hinted")])
def test_few_groups_under_more_than_six_columns(ncols, groups, shape, monkeypatch):
    """Few groups under MORE than six plain 8-byte columns: the program is cut into parts of up to six columns (make_parts), each through
    agg_hotn_kernel, joined by key -- instead of the interpreted scan over all columns at once.  Bit-exact against the oracle (float64 /
    int64 columns, sums, averages, counts, extremes, a predicate on another column; several batches; a caller's group count)."""
    from oracle import oracle as O
    from vinum_amd import _lib as L
    monkeypatch.setenv("VNM_AGG_ESTIMATE_MIN_ROWS", "100000")
    rng = np.random.default_rng(ncols * 41 + groups)
    n = 420_000
    cols = {"k": pa.array(rng.integers(0, groups, n).astype(np.int64) * 13 + 5)}
    kinds = [O.SUM, O.AVG, O.COUNT, O.MAX, O.SUM, O.MIN]
    funcs = []
    for i in range(ncols):
        if i % 4 == 3:
            cols[f"c{i}"] = pa.array(rng.integers(-2**40, 2**40, n).astype(np.int64))
        else:
            cols[f"c{i}"] = pa.array(rng.integers(-2**13, 2**13, n).astype(np.float64) / 32.0)
        funcs.append((kinds[i % 6], f"c{i}", f"f{i}"))
    funcs.append((O.COUNT_STAR, "", "n"))
    cols["p"] = pa.array(rng.integers(0, 2**12, n).astype(np.float64) / 64.0)
    t = pa.table(cols)
    bl = util.sliced_batches(t, 210_000 if shape == "two_batches" else n)
    pred = ("p", ">", 20.0)
    L.lib().vnm_set_profiling(1)
    got = gpu_aggregate(O.SINGLE, ["k"], ["k"], funcs, bl, predicate=pred, expected_groups=groups if shape == "hinted" else 0)
    joins = _launches(b"agg_split_join")
    L.lib().vnm_set_profiling(0)
    assert joins == 1, joins
    util.assert_agg_equal(got, _oracle(O.SINGLE, ["k"], funcs, bl, pred), funcs, ["k"], exact_float_inputs=tuple(f"c{i}" for i in range(ncols)),
                          what=f"few groups, {ncols} columns, G={groups} {shape}")
