"""Full-size (BASELINE.json sizes: 1e9 rows) property checks of the hot path on one MI355X.

The oracle finishes in seconds only up to ~1e6 rows, so at full size the HIP path is checked through
size-independent properties evaluated ON THE DEVICE with plain torch ops (sort / unique / index_add -- no code of
ours): survivors are conserved, sums are conserved exactly (the benchmark data is quantised, every partial sum is an
exactly representable double), the number of groups equals torch.unique, and every group of a 1/64 key-hash subsample
bit-equals a torch reference.  Same generators as bench.py.
"""
import ctypes

import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu

N_FULL = 1_000_000_000


def _torch():
    import torch
    return torch


class _View:
    """raw device pointer -> torch tensor (zero copy)"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _as_tensor(ptr, n, typestr="<i8"):
    torch = _torch()
    return torch.as_tensor(_View(ptr, n, typestr), device="cuda")


def _gen(n, groups, seed=1):
    import bench
    torch = _torch()
    return bench.gen_data(torch, n, groups, seed, torch.device("cuda", 0))


def _free():
    torch = _torch()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def _sum_words(agg):
    """(count word, sum word, compensation word or None) of a `sum(v), avg(v)` program, from the library's own plan."""
    from vinum_amd import _lib as L
    lay = agg.word_layout()
    w_cnt = next(w for kind, col, w in lay["ops"] if kind == 1)
    w_sum = next(w for kind, col, w in lay["ops"] if kind == 2)
    w_lo = w_sum + 1 if lay["merge"][w_sum] == L.M_ADD_F64C else None
    return w_cnt, w_sum, w_lo


@pytest.mark.parametrize("groups,hint", [(7, 0), (100_000_000, 0), (100_000_000, 100_000_000), (1_000_000, 0)],
                         ids=["G7", "G1e8-hintless", "G1e8-hinted", "G1e6-hintless"])
def test_groupby_1e9_rows(groups, hint):
    """BASELINE configs[2] at full size: SELECT k, sum(v), avg(v) WHERE v > X GROUP BY k; N = 1e9."""
    torch = _torch()
    import bench
    from vinum_amd import _lib as L
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    n = N_FULL
    k, v = _gen(n, groups)
    x = bench.threshold_for(0.5)
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())],
                              expected_groups=hint)
    agg.set_predicate(">", x)
    kc, vc = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
    agg.next([kc], [vc, vc], pred=vc, nrows=n)
    ng = agg.finish()
    torch.cuda.synchronize()
    w_cnt, w_sum, w_lo = _sum_words(agg)
    kptrs, aptrs = agg.dense_ptrs()
    rk = _as_tensor(kptrs[0], ng)
    rnull = _as_tensor(kptrs[1], ng)
    rcnt = _as_tensor(aptrs[w_cnt], ng)
    rsum = _as_tensor(aptrs[w_sum], ng, "<f8")
    if w_lo is not None:
        rlo = _as_tensor(aptrs[w_lo], ng, "<f8")
        assert int((rlo != 0).sum()) == 0, "quantised data: every add is exact, the compensation words must be zero"

    # 1. survivors are conserved
    keep = v > x
    survivors = int(keep.sum())
    assert int(rcnt.sum()) == survivors
    assert int((rcnt <= 0).sum()) == 0
    assert int(rnull.sum()) == 0
    # 2. the total is conserved exactly: v = j / 128, so 128 * sum is an exact integer below 2^53
    tot_ref = int((v * 128.0).to(torch.int64)[keep].sum())
    scaled = rsum * 128.0
    assert bool((scaled == scaled.round()).all())
    assert int(scaled.to(torch.int64).sum()) == tot_ref
    # 3. group count == torch.unique over the surviving keys; keys are distinct
    ks = k[keep]
    uniq = torch.unique(ks)
    assert ng == uniq.numel()
    srt, order = torch.sort(rk)
    assert bool(torch.equal(srt, uniq))
    del uniq
    # 4. every group of a 1/64 key-hash subsample bit-equals a torch reference (sum, count) -- and avg through the
    #    library's finaliser below
    def sel(keys):
        return (((keys * -7046029254386353131) >> 58) & 63) == 0
    vs = v[keep]
    m = sel(ks)
    sub_k, sub_v = ks[m], vs[m]
    del ks, vs, m, keep
    uk, inv = torch.unique(sub_k, return_inverse=True)
    ref_sum = torch.zeros(uk.numel(), dtype=torch.float64, device=k.device).index_add_(0, inv, sub_v)
    ref_cnt = torch.bincount(inv, minlength=uk.numel())
    ms = sel(srt)
    got_k = srt[ms]
    got_sum = rsum[order][ms]
    got_cnt = rcnt[order][ms]
    assert bool(torch.equal(got_k, uk))
    assert bool(torch.equal(got_sum.view(torch.int64), ref_sum.view(torch.int64)))   # bit-exact
    assert bool(torch.equal(got_cnt, ref_cnt))
    # 5. the finalised result columns (key, sum, avg: what BaseAggregate::Result returns) on the device
    cols = agg.result_device()
    assert [c.length for c in cols] == [ng, ng, ng]
    fk = _as_tensor(cols[0].values_ptr, ng)
    fs = _as_tensor(cols[1].values_ptr, ng, "<f8")
    fa = _as_tensor(cols[2].values_ptr, ng, "<f8")
    assert all(c.validity_ptr is None for c in cols)
    assert bool(torch.equal(fk, rk))
    assert bool(torch.equal(fs.view(torch.int64), rsum.view(torch.int64)))
    assert bool(torch.equal(fa.view(torch.int64), (rsum / rcnt.to(torch.float64)).view(torch.int64)))
    agg.close()
    del k, v
    _free()


@pytest.mark.parametrize("groups,mode", [(100_000_000, "one_batch"), (1_000_000, "one_batch"), (7, "one_batch"),
                                         (1_000_000, "stream"), (7, "stream"), (100_000_000, "stream")],
                         ids=["G1e8", "G1e6", "G7", "G1e6-stream", "G7-stream", "G1e8-stream"])
def test_groupby_1e9_rows_result_columns(groups, mode):
    """The BENCHED path at full size: next() -> result_device() with NO finish() first, i.e. the deferred final pass
    writing the result columns itself (DF_COLS) -- checked on those columns directly, against plain torch ops: survivors conserved,
    totals conserved exactly, group count and keys == torch.unique, a 1/64 key-hash subsample bit-equal (sum, and avg = sum / count).
    mode "stream": the same rows as 59 record batches of 2^24 rows through an operator in stream mode (vnm_agg_set_async: the
    batches go to the device as the segments of one launch) -- bench.py's configs[3] leg."""
    torch = _torch()
    import bench
    from vinum_amd import _lib as L
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    B = 1 << 24
    n = N_FULL if mode == "one_batch" else 59 * B
    k, v = _gen(n, groups)
    x = bench.threshold_for(0.5)
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64()), (L.COUNT_STAR, None, None)],
                              stream_mode=(mode == "stream"))
    agg.set_predicate(">", x)
    if mode == "one_batch":
        kc, vc = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
        agg.next([kc], [vc, vc, None], pred=vc, nrows=n)
    else:
        for i in range(n // B):
            kc, vc = DeviceColumn.from_torch(k[i * B:(i + 1) * B]), DeviceColumn.from_torch(v[i * B:(i + 1) * B])
            agg.next([kc], [vc, vc, None], pred=vc, nrows=B)
    cols = agg.result_device()           # <- the step of bench.py ends here
    torch.cuda.synchronize()
    ng = agg.result_rows
    assert [c.length for c in cols] == [ng] * 4 and all(c.validity_ptr is None for c in cols)
    rk = _as_tensor(cols[0].values_ptr, ng)
    rsum = _as_tensor(cols[1].values_ptr, ng, "<f8")
    ravg = _as_tensor(cols[2].values_ptr, ng, "<f8")
    rcnt = _as_tensor(cols[3].values_ptr, ng)
    keep = v > x
    assert int(rcnt.sum()) == int(keep.sum()) and int((rcnt <= 0).sum()) == 0                       # 1. survivors
    scaled = rsum * 128.0
    assert bool((scaled == scaled.round()).all())
    assert int(scaled.to(torch.int64).sum()) == int((v * 128.0).to(torch.int64)[keep].sum())        # 2. exact total
    ks = k[keep]
    uniq = torch.unique(ks)
    assert ng == uniq.numel()                                                                        # 3. groups and keys
    srt, order = torch.sort(rk)
    assert bool(torch.equal(srt, uniq))
    del uniq

    def sel(keys):
        return (((keys * -7046029254386353131) >> 58) & 63) == 0
    vs = v[keep]
    m = sel(ks)
    sub_k, sub_v = ks[m], vs[m]
    del ks, vs, m, keep
    uk, inv = torch.unique(sub_k, return_inverse=True)
    ref_sum = torch.zeros(uk.numel(), dtype=torch.float64, device=k.device).index_add_(0, inv, sub_v)
    ref_cnt = torch.bincount(inv, minlength=uk.numel())
    ms = sel(srt)
    assert bool(torch.equal(srt[ms], uk))                                                           # 4. subsample, bit-exact
    assert bool(torch.equal(rsum[order][ms].view(torch.int64), ref_sum.view(torch.int64)))
    assert bool(torch.equal(rcnt[order][ms], ref_cnt))
    assert bool(torch.equal(ravg.view(torch.int64), (rsum / rcnt.to(torch.float64)).view(torch.int64)))   # 5. avg = sum / count, every group
    agg.close()
    del k, v
    _free()


@pytest.mark.parametrize("groups", [1_000, 100_000, 100_000_000], ids=["G1e3", "G1e5", "G1e8"])
def test_generic_programs_1e9_rows(groups):
    """The generic dense paths at full size (whole-table LDS scan, split final pass, two scatter levels):
    SELECT k, count(*) GROUP BY k and SELECT k, min(v), max(v), count(*) WHERE v > X GROUP BY k over 1e9 rows, finalised on
    the device and compared with torch.bincount / scatter_reduce for EVERY group."""
    torch = _torch()
    import bench
    from vinum_amd import _lib as L
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    n = N_FULL
    k, v = _gen(n, groups)
    kc, vc = DeviceColumn.from_torch(k), DeviceColumn.from_torch(v)
    # ---- count(*): the entries are bare key codes
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.COUNT_STAR, None, None)])
    agg.next([kc], [None], nrows=n)
    ng = agg.finish()
    cols = agg.result_device()
    fk = _as_tensor(cols[0].values_ptr, ng)
    fc = _as_tensor(cols[1].values_ptr, ng)
    ref = torch.bincount(k, minlength=groups)
    assert ng == int((ref > 0).sum())
    assert int(fc.sum()) == n
    assert bool(torch.equal(ref[fk], fc)), "count(*) differs from torch.bincount for some group"
    assert torch.unique(fk).numel() == ng
    agg.close()
    del ref, fk, fc, cols
    _free()
    # ---- min, max, count(*) with the predicate on the value column
    x = bench.threshold_for(0.5)
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.MIN, 1, pa.float64()), (L.MAX, 1, pa.float64()), (L.COUNT_STAR, None, None)])
    agg.set_predicate(">", x)
    agg.next([kc], [vc, vc, None], pred=vc, nrows=n)
    ng = agg.finish()
    cols = agg.result_device()
    fk = _as_tensor(cols[0].values_ptr, ng)
    fmin = _as_tensor(cols[1].values_ptr, ng, "<f8")
    fmax = _as_tensor(cols[2].values_ptr, ng, "<f8")
    fc = _as_tensor(cols[3].values_ptr, ng)
    keep = v > x
    ks, vs = k[keep], v[keep]
    del keep
    rc = torch.bincount(ks, minlength=groups)
    assert ng == int((rc > 0).sum())
    assert bool(torch.equal(rc[fk], fc))
    del rc
    rmin = torch.full((groups,), float("inf"), dtype=torch.float64, device=k.device).scatter_reduce_(0, ks, vs, "amin")
    assert bool(torch.equal(rmin[fk], fmin)), "min(v) differs from scatter_reduce(amin) for some group"
    del rmin
    rmax = torch.full((groups,), float("-inf"), dtype=torch.float64, device=k.device).scatter_reduce_(0, ks, vs, "amax")
    assert bool(torch.equal(rmax[fk], fmax)), "max(v) differs from scatter_reduce(amax) for some group"
    agg.close()
    del k, v, ks, vs, rmax
    _free()


def test_filter_1e9_rows():
    """BASELINE configs[1] at full size: WHERE v > X -> compacted column; order preserving, so the output must EQUAL
    torch's boolean indexing."""
    torch = _torch()
    import bench
    from vinum_amd import _lib as L
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    n = N_FULL
    _, v = _gen(n, 7)
    for sel in (0.5, 0.01, 0.99):
        x = bench.threshold_for(sel)
        outs, cnt = ops.filter_cmp(DeviceColumn.from_torch(v), ">", float(x), [DeviceColumn.from_torch(v)])
        torch.cuda.synchronize()
        ref = v[v > x]
        assert cnt == ref.numel()
        got = _as_tensor(outs[0].values_ptr, cnt, "<f8")
        assert bool(torch.equal(got.view(torch.int64), ref.view(torch.int64)))
        del outs, ref, got
        _free()
    del v
    _free()


@pytest.mark.parametrize("limit", [10, 100_000, 0], ids=["K10", "K1e5", "fullsort"])
def test_order_by_1e9_rows(limit):
    """BASELINE configs[4]: ORDER BY v DESC [LIMIT K] over 1e9 fp64 rows (0.1 % NaN): values non-increasing with NaN
    last, top-K equals torch.topk, a full sort is a permutation, and ties keep row order (stable)."""
    torch = _torch()
    from vinum_amd import _lib as L
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    n = N_FULL
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    v = torch.randn(n, device="cuda", dtype=torch.float64, generator=g) * 3.0 + 11.0
    # duplicates (ties) and NaNs
    v[::1000] = 12.5
    v[7::1000] = float("nan")
    idx = ops.sort_indices([DeviceColumn.from_torch(v)], [L.DESC], limit=limit)
    torch.cuda.synchronize()
    m = limit if limit else n
    ids = _as_tensor(idx.ptr, m)
    assert int(ids.min()) >= 0 and int(ids.max()) < n
    got = v[ids]
    if limit:
        clean = torch.where(torch.isnan(v), torch.full_like(v, float("-inf")), v)
        ref = torch.topk(clean, limit, sorted=True).values
        assert bool(torch.equal(got.view(torch.int64), ref.view(torch.int64)))
        del clean, ref
    else:
        nn = int(torch.isnan(v).sum())
        body = got[: n - nn]
        assert not bool(torch.isnan(body).any())
        assert bool(torch.isnan(got[n - nn:]).all())          # NaN after every number (Arrow SortIndices)
        assert bool((body[1:] <= body[:-1]).all())
        flag = torch.zeros(n, dtype=torch.uint8, device="cuda")
        flag[ids] = 1
        assert bool(flag.all())                                 # a permutation
        del flag, body
    # stability: equal keys keep ascending row ids
    same = got[1:] == got[:-1]
    assert bool((ids[1:][same] > ids[:-1][same]).all())
    del v, got, ids, idx
    _free()


def test_order_by_1e9_rows_with_null_keys():
    """configs[4]'s NULL variant: ORDER BY v DESC over 1e9 fp64 rows, 0.1 % NaN and 0.1 % NULL keys: numbers non-increasing, then
    the NaNs, then the NULL rows (Arrow SortIndices: for either direction); a permutation; ties, NaNs and NULLs in row order."""
    torch = _torch()
    from vinum_amd import _lib as L
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    n = N_FULL
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    v = torch.randn(n, device="cuda", dtype=torch.float64, generator=g) * 3.0 + 11.0
    v[::1000] = 12.5
    v[7::1000] = float("nan")
    bits = torch.full(((n + 7) // 8,), 255, dtype=torch.uint8, device="cuda")
    bits[5::125] = 0b11110111                       # row 8 * (5 + 125 j) + 3 is NULL: 0.1 % of the rows
    idx = ops.sort_indices([DeviceColumn.from_torch(v, validity=bits)], [L.DESC])
    torch.cuda.synchronize()
    ids = _as_tensor(idx.ptr, n)
    assert int(ids.min()) >= 0 and int(ids.max()) < n
    is_null = ((bits[ids >> 3] >> (ids & 7).to(torch.uint8)) & 1) == 0
    n_null = int(is_null.sum())
    assert n_null == (bits.numel() - 5 + 124) // 125
    assert bool(is_null[n - n_null:].all())                    # the NULL rows last ...
    tail = ids[n - n_null:]
    assert bool((tail[1:] > tail[:-1]).all())                  # ... in row order
    got = v[ids[: n - n_null]]
    nn = int(torch.isnan(got).sum())
    body = got[: n - n_null - nn]
    assert not bool(torch.isnan(body).any()) and bool(torch.isnan(got[n - n_null - nn:]).all())
    assert bool((body[1:] <= body[:-1]).all())
    flag = torch.zeros(n, dtype=torch.uint8, device="cuda")
    flag[ids] = 1
    assert bool(flag.all())
    same = got[1:] == got[:-1]
    head = ids[: n - n_null]
    assert bool((head[1:][same] > head[:-1][same]).all())
    del v, got, ids, idx, flag, bits
    _free()


def test_projection_1e9_rows():
    """BASELINE configs[4]: `v*2+1, v-a, a*b` over 1e9 rows, one fused kernel; IEEE double results bit-equal torch's
    (separate multiply and add: NumPy does not contract either)."""
    torch = _torch()
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    n = N_FULL
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    v = torch.randn(n, device="cuda", dtype=torch.float64, generator=g) * 3.0 + 11.0
    a = torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
    b = torch.rand(n, device="cuda", dtype=torch.float64, generator=g)
    cols = {"v": DeviceColumn.from_torch(v), "a": DeviceColumn.from_torch(a), "b": DeviceColumn.from_torch(b)}
    outs = ops.project_many([("add", ("mul", "v", 2), 1), ("sub", "v", "a"), ("mul", "a", "b")], cols, length=n)
    torch.cuda.synchronize()
    refs = [lambda: (v * 2.0) + 1.0, lambda: v - a, lambda: a * b]
    for o, r in zip(outs, refs):
        got = _as_tensor(o.values_ptr, n, "<f8")
        ref = r()
        assert bool(torch.equal(got.view(torch.int64), ref.view(torch.int64)))
        del ref, got
    del outs, v, a, b
    _free()


def test_ordered_min_max_1e9_rows_against_torch():
    """Round 5: MinMaxFunc's row-order rule (agg_funcs.h:188-201) at full size -- 1e9 rows in four record batches, G = 1e6, a thousand
    NaNs and a thousand -0.0 among values that hold +0.0 anyway.  MIN = NaN iff the group's first row is NaN, else the smallest
    number, the EARLIEST of tied zeros; MAX = the largest value after the group's LAST NaN (NaN if that is its last row), the LATEST of
    tied zeros.  The expectation is built from global row positions with plain torch scatter_reduce; every group is compared bit for
    bit (the first batch is clean: the operator switches to the ordered side table in the second)."""
    torch = _torch()
    from vinum_amd import _lib as L
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    n, groups = N_FULL, 1_000_000
    k, v = _gen(n, groups)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    idx = torch.randint(n // 4, n, (2000,), device=k.device, generator=gen)
    v[idx[:1000]] = float("nan")
    v[idx[1000:]] = -0.0
    idx2 = torch.randint(0, n, (200_000,), device=k.device, generator=gen)       # make zero the minimum of many groups' rows a tie
    v[idx2[idx2 >= n // 4]] = 0.0
    agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.MIN, 1, pa.float64()), (L.MAX, 1, pa.float64()), (L.COUNT_STAR, None, None)])
    keep = []
    for b in range(4):
        lo, hi = b * (n // 4), (b + 1) * (n // 4)
        kc, vc = DeviceColumn.from_torch(k[lo:hi]), DeviceColumn.from_torch(v[lo:hi])
        keep.append((kc, vc))
        agg.next([kc], [vc, vc, None], nrows=hi - lo)
    ng = agg.finish()
    cols = agg.result_device()
    fk = _as_tensor(cols[0].values_ptr, ng)
    fmin = _as_tensor(cols[1].values_ptr, ng)
    fmax = _as_tensor(cols[2].values_ptr, ng)
    fc = _as_tensor(cols[3].values_ptr, ng)
    assert ng == groups and int(fc.sum()) == n
    dev = k.device
    pos = torch.arange(n, device=dev)
    isn = torch.isnan(v)
    inf = float("inf")
    # ---- MIN
    first = torch.full((groups,), n, dtype=torch.int64, device=dev).scatter_reduce_(0, k, pos, "amin")
    first_nan = isn[first]
    rmin = torch.full((groups,), inf, dtype=torch.float64, device=dev).scatter_reduce_(0, k, torch.where(isn, inf, v), "amin")
    zero = v == 0
    fz = torch.full((groups,), n, dtype=torch.int64, device=dev).scatter_reduce_(0, k, torch.where(zero, pos, n), "amin")
    zsel = (rmin == 0) & (fz < n)
    rmin = torch.where(zsel, v[fz.clamp(max=n - 1)], rmin)              # the earliest zero, with its sign
    rmin = torch.where(first_nan, float("nan"), rmin)
    assert bool(torch.equal(rmin.view(torch.int64)[fk], fmin)), "min(v) differs from the row-order rule for some group"
    del first, first_nan, rmin, fz, zsel
    # ---- MAX
    lastnan = torch.full((groups,), -1, dtype=torch.int64, device=dev).scatter_reduce_(0, k, torch.where(isn, pos, -1), "amax")
    after = pos > lastnan[k]
    rmax = torch.full((groups,), -inf, dtype=torch.float64, device=dev).scatter_reduce_(0, k, torch.where(after, v, -inf), "amax")
    lz = torch.full((groups,), -1, dtype=torch.int64, device=dev).scatter_reduce_(0, k, torch.where(after & zero, pos, -1), "amax")
    zsel = (rmax == 0) & (lz >= 0)
    rmax = torch.where(zsel, v[lz.clamp(min=0)], rmax)                 # the latest zero, with its sign
    rmax = torch.where(rmax == -inf, float("nan"), rmax)               # nothing after the last NaN: it was the group's last row
    assert bool(torch.equal(rmax.view(torch.int64)[fk], fmax)), "max(v) differs from the row-order rule for some group"
    assert int(torch.isnan(rmax).sum()) + int(isn.sum()) > 0
    agg.close()
    del k, v, pos, isn, after, rmax, lz, zero, lastnan
    _free()
