"""Multi-GPU group-by: batch-sharded partial aggregates + ONE key-partitioned exchange.

The reference has no distributed code at all; what makes sharding legal is structural: operators consume
independent RecordBatches and every aggregate state is a commutative monoid per group (base_aggregate.cpp:23-45
only ever Init()s or Update()s a group's state).  So each rank aggregates its own batches into a dense run of
partial groups (key words + 64-bit accumulator words that merge by add / min / max), then

    owner(key) = mix(key words) mod P
    all_to_all of the runs, bucketed by owner           (RCCL over xGMI: the full mesh, not a ring)
    the owner merges what it received                   (same merge kinds as the LDS / HBM tables)

and the final result is the concatenation of the owners' shards (row order is unspecified in the reference
anyway, robin_hood iteration order).  One process per GPU; torch.distributed is only the transport.

The merge itself is a device operation (DeviceAggregate.merge); this module takes it as a callable so that the
CPU tests (gloo, world_size 2) can exercise the ownership / exchange logic with a host stand-in.
"""
import torch
import torch.distributed as dist

_MIX = -7046029254386353131  # 0x9E3779B97F4A7C15 as int64


def _all_to_all_single(out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
    """dist.all_to_all_single -- RCCL over xGMI with the `nccl` backend.  With `gloo` (CPU tests; two ranks sharing ONE GPU in
    tests/test_gpu_bench_check.py, where RCCL refuses two ranks on a device) device tensors are staged through the host."""
    if out.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
        out.copy_(o)
        return
    dist.all_to_all_single(out, inp, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)


def owner_of(key_words, world: int) -> torch.Tensor:
    """Owner rank per partial group.  key_words: list of int64 tensors (n_keys value words + null-mask word)."""
    h = torch.zeros_like(key_words[0])
    for w in key_words:
        h = (h ^ w) * _MIX            # wraps in int64: a multiplicative hash per word
        h = h ^ (h >> 29)
    return ((h >> 17) & 0x7FFFFFFF) % world


def bucket_by_owner(words, n_key_words: int, world: int):
    """words: list of int64 tensors of equal length (key words first).  Returns (rows [n, n_words] grouped by
    owner, counts [world])."""
    if len(words[0]) == 0:
        return torch.empty((0, len(words)), dtype=torch.int64, device=words[0].device), \
            torch.zeros(world, dtype=torch.int64, device=words[0].device)
    owner = owner_of(words[:n_key_words], world)
    order = torch.argsort(owner, stable=True)
    counts = torch.bincount(owner, minlength=world)
    send = torch.stack([w[order] for w in words], dim=1).contiguous()
    return send, counts


_MAX_ELEMS_PER_PEER = 1 << 27  # 1 GiB of int64 per peer per all_to_all round (keeps byte counts far below 2^31..2^32)


def exchange(send: torch.Tensor, counts: torch.Tensor, group=None) -> torch.Tensor:
    """all_to_all of variable-size row blocks (grouped by destination rank); returns the rows this rank owns,
    grouped by source rank.  Large exchanges run in several rounds of at most 1 GiB per peer."""
    world = dist.get_world_size(group)
    ncol = send.shape[1]
    recv_counts = torch.empty_like(counts)
    _all_to_all_single(recv_counts, counts, group=group)
    sc, rc = counts.tolist(), recv_counts.tolist()
    assert world == len(sc)
    rows_per_round = max(1, _MAX_ELEMS_PER_PEER // ncol)
    biggest = torch.tensor([max(sc + rc)], dtype=torch.int64, device=send.device)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)
    rounds = max(1, -(-int(biggest.item()) // rows_per_round))
    recv = torch.empty((sum(rc), ncol), dtype=send.dtype, device=send.device)
    s_off = [0] * world
    r_off = [0] * world
    for o in range(1, world):
        s_off[o] = s_off[o - 1] + sc[o - 1]
        r_off[o] = r_off[o - 1] + rc[o - 1]
    if rounds == 1:
        # flat 1-D buffers with element-count splits: the form both RCCL and gloo accept
        _all_to_all_single(recv.view(-1), send.view(-1), output_split_sizes=[c * ncol for c in rc],
                               input_split_sizes=[c * ncol for c in sc], group=group)
        return recv
    for r in range(rounds):
        lo = r * rows_per_round
        s_n = [max(0, min(c - lo, rows_per_round)) for c in sc]
        r_n = [max(0, min(c - lo, rows_per_round)) for c in rc]
        s_buf = torch.cat([send[s_off[o] + lo: s_off[o] + lo + s_n[o]] for o in range(world)]).contiguous()
        r_buf = torch.empty((sum(r_n), ncol), dtype=send.dtype, device=send.device)
        _all_to_all_single(r_buf.view(-1), s_buf.view(-1), output_split_sizes=[c * ncol for c in r_n],
                               input_split_sizes=[c * ncol for c in s_n], group=group)
        pos = 0
        for o in range(world):
            recv[r_off[o] + lo: r_off[o] + lo + r_n[o]] = r_buf[pos: pos + r_n[o]]
            pos += r_n[o]
    return recv


def exchange_bucketed(send: torch.Tensor, counts, merge, group=None):
    """send: rows [n, n_words] already grouped by owner (vnm_agg_bucket_by_owner), counts: rows per owner."""
    c = torch.tensor(list(counts), dtype=torch.int64, device=send.device)
    recv = exchange(send, c, group)
    return merge(recv)


def exchange_partials(words, n_key_words: int, merge, group=None):
    """words: this rank's dense run (list of int64 tensors).  merge(list_of_word_tensors) -> anything: called with
    the rows this rank owns (possibly from several ranks, duplicates across ranks included)."""
    world = dist.get_world_size(group)
    send, counts = bucket_by_owner(words, n_key_words, world)
    recv = exchange(send, counts, group)
    cols = [recv[:, i].contiguous() for i in range(recv.shape[1])]
    return merge(cols)


def first_partition(owner: int, nfin: int, world: int) -> int:
    """First hash partition owned by `owner` (owner(f) = f * world // nfin)."""
    return -(-owner * nfin // world)


def exchange_partition_aligned(agg, make_merged, device, group=None):
    """Large-G exchange: every rank holds a partition-structured run with the SAME number of hash partitions F.
    Rows travel in partition order, the per-partition row counts travel with them, and each owner merges its
    partitions in LDS (vnm_agg_merge_partitioned) -- no HBM atomics on the receiving side.
    Returns the merged DeviceAggregate, or None when the ranks do not all hold such a run (caller falls back to
    the owner-bucketed exchange)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    kw0, aw0 = agg.layout()
    nfin = agg.run_partitions()   # 0 unless every word add-merges and the group fits the LDS merge table
    t = torch.tensor([nfin, -nfin], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    if int(t[0]) <= 0 or int(t[0]) != -int(t[1]) or nfin < world:
        return None
    n = agg.finish()
    kw, aw = agg.layout()
    send = torch.empty((max(n, 1), kw + aw), dtype=torch.int64, device=device)
    pc = torch.empty(nfin, dtype=torch.int32, device=device)
    counts = agg.run_reorder(world, send.data_ptr(), pc.data_ptr())
    recv = exchange(send[:n], torch.tensor(counts, dtype=torch.int64, device=device), group)
    # per-partition counts: owner o gets pc[first(o):first(o+1)] from every rank
    bounds = [first_partition(o, nfin, world) for o in range(world + 1)]
    nlocal = bounds[rank + 1] - bounds[rank]
    pc_recv = torch.empty(world * nlocal, dtype=torch.int32, device=device)
    _all_to_all_single(pc_recv, pc, output_split_sizes=[nlocal] * world,
                           input_split_sizes=[bounds[o + 1] - bounds[o] for o in range(world)], group=group)
    # row offsets of the source blocks inside recv
    rc = torch.empty(world, dtype=torch.int64, device=device)
    _all_to_all_single(rc, torch.tensor(counts, dtype=torch.int64, device=device), group=group)
    offs = [0]
    for c in rc.tolist():
        offs.append(offs[-1] + c)
    merged = make_merged()
    ok = merged.merge_partitioned(world, nlocal, recv.data_ptr(), offs, pc_recv.data_ptr())
    # A merged partition that outgrows the LDS table (ranks with disjoint key sets) is a per-rank event, but what
    # follows must stay collective-free AND consistent: the rows are already here, so the rank that could not merge
    # them partition by partition merges the same rows through its HBM table instead -- no second exchange.
    if not ok:
        merged.merge_rows(int(recv.shape[0]), recv.data_ptr())
    merged._keep = (recv, pc_recv)
    return merged


def agree_on_group_count(agg, key, nrows, device, group=None, stream=None):
    """Every rank estimates its batch's group count with the operator's own estimator, the ranks agree on the MAXIMUM (one
    8-byte all_reduce) and hand it to the operator as its hint.  The number of hash partitions follows the hint, so without
    this two ranks whose estimates fall on either side of a power-of-two boundary (G = 1e8: 1.2e8 / 900 = 133 k vs the boundary
    at 131 072) would cut their results differently and the partition-aligned exchange would fall back to the bucketed one."""
    est = agg.estimate_groups(key, nrows, stream=stream)
    t = torch.tensor([est], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    est = int(t.item())
    if est > 0:
        agg.set_hint(est)
    return est


# ---- large results of the dense-key path: direct-addressed tables, exchanged as they are ------------------------------------
class _RawView:
    """A raw device pointer as a __cuda_array_interface__ object (zero copy into torch)."""

    def __init__(self, ptr, n, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _u64_to_ordered_i64(u: int) -> int:
    """Order-preserving map of an unsigned 64-bit value onto int64 (torch has no uint64 reductions)."""
    u ^= 1 << 63
    return u - (1 << 64) if u >= (1 << 63) else u


def _ordered_i64_to_u64(v: int) -> int:
    return (v + (1 << 64) if v < 0 else v) ^ (1 << 63)


def agree_on_dense_range(agg, key, nrows, device, group=None, stream=None):
    """Every rank samples the key range of its batch (vnm_agg_dense_range); the ranks agree on [MIN lo, MAX hi] with one
    16-byte all_gather per rank and hand it to their operators (vnm_agg_set_dense_range): all of them then derive the SAME
    code map, so the direct-addressed tables of the dense path's final pass are slot-compatible (exchange_dense_tables).
    A rank without such a range (empty batch, another key type) switches it off for everybody.  Returns (lo, hi) or None."""
    world = dist.get_world_size(group)
    lo, hi = agg.dense_range(key, nrows, stream=stream)
    ok = 1 if lo <= hi else 0
    t = torch.tensor([ok, _u64_to_ordered_i64(lo), _u64_to_ordered_i64(hi)], dtype=torch.int64, device=device)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t, group=group)
    rows = [x.tolist() for x in allt]
    if not all(r[0] for r in rows):
        agg.set_dense_range(1, 0)
        return None
    lo = _ordered_i64_to_u64(min(r[1] for r in rows))
    hi = _ordered_i64_to_u64(max(r[2] for r in rows))
    agg.set_dense_range(lo, hi)
    return lo, hi


def agree_on_groups_and_range(agg, key, nrows, device, group=None, stream=None, estimate=True):
    """agree_on_group_count + agree_on_dense_range with ONE collective (round 4): every rank contributes (estimate, range ok, lo, hi)
    to one all_gather -- a step of a stream pays the latency of an agreement once, not twice (two collectives plus their host
    round trips were ~0.3 ms of a 7.4 ms step at G = 1e6).  estimate = False: the caller gave a hint, only the range is agreed on.
    Returns (estimate or 0, (lo, hi) or None)."""
    world = dist.get_world_size(group)
    est = agg.estimate_groups(key, nrows, stream=stream) if estimate else 0
    lo, hi = agg.dense_range(key, nrows, stream=stream)
    ok = 1 if lo <= hi else 0
    t = torch.tensor([est, ok, _u64_to_ordered_i64(lo), _u64_to_ordered_i64(hi)], dtype=torch.int64, device=device)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t, group=group)
    rows = [x.tolist() for x in allt]
    est = max(r[0] for r in rows)
    if estimate and est > 0:
        agg.set_hint(est)
    if not all(r[1] for r in rows):
        agg.set_dense_range(1, 0)
        return est, None
    lo = _ordered_i64_to_u64(min(r[2] for r in rows))
    hi = _ordered_i64_to_u64(max(r[3] for r in rows))
    agg.set_dense_range(lo, hi)
    return est, (lo, hi)


def table_bounds(nslots: int, world: int):
    """Owner o holds the slots (= scrambled key codes) [bounds[o], bounds[o + 1])."""
    return [o * nslots // world for o in range(world + 1)]


def exchange_dense_tables(table, geometry, merge_slices, group=None):
    """table: this rank's direct-addressed final-pass tables as an int64 tensor [nslots, 2] (16-byte slots {sum f64, lo f32,
    count u32}; vnm_agg_dense_table) or None when its batch did not take that path; geometry: the code map's
    (range start, bits, multiplier, sign).  When EVERY rank holds a table of the same geometry: ONE all_to_all with equal,
    known splits (owner o gets the slot range table_bounds()[o : o + 2] of every rank -- nothing is bucketed, counted or
    re-ordered, no sizes are exchanged) and merge_slices(recv [world * nloc, 2], code0, nloc) adds the world slices up
    (vnm_agg_merge_dense_tables).  Returns its result, or None when some rank has no table / another geometry (all ranks
    agree on that: one all_gather of the 5-word signature), in which case the caller uses another exchange."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = table.device if table is not None else None
    if dev is None:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    sig = [1 if table is not None else 0] + ([_u64_to_ordered_i64(int(g)) for g in geometry] if table is not None else [0, 0, 0, 0])
    t = torch.tensor(sig, dtype=torch.int64, device=dev)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t, group=group)
    rows = [x.tolist() for x in allt]
    if not all(r[0] == 1 and r == rows[0] for r in rows):
        return None
    nslots = int(table.shape[0])
    bounds = table_bounds(nslots, world)
    nloc = bounds[rank + 1] - bounds[rank]
    recv = torch.empty((world * nloc, 2), dtype=torch.int64, device=dev)
    biggest = max(bounds[o + 1] - bounds[o] for o in range(world))
    rounds = max(1, -(-(biggest * 2) // _MAX_ELEMS_PER_PEER))
    if rounds == 1:
        _all_to_all_single(recv.view(-1), table.view(-1), output_split_sizes=[nloc * 2] * world,
                               input_split_sizes=[(bounds[o + 1] - bounds[o]) * 2 for o in range(world)], group=group)
        return merge_slices(recv, bounds[rank], nloc)
    # slices beyond 1 GiB per peer (one or two ranks with a 2^27-slot table): several rounds over sub-ranges of every owner's
    # slots, so that no single transfer comes near the 2^31-byte counts of the transport
    step = -(-biggest // rounds)
    for r in range(rounds):
        s_lo = [min(bounds[o] + r * step, bounds[o + 1]) for o in range(world)]
        s_hi = [min(bounds[o] + (r + 1) * step, bounds[o + 1]) for o in range(world)]
        r_lo, r_hi = min(r * step, nloc), min((r + 1) * step, nloc)
        send = table[s_lo[0]:s_hi[0]] if world == 1 else torch.cat([table[a:b] for a, b in zip(s_lo, s_hi)])
        rbuf = recv[r_lo:r_hi] if world == 1 else torch.empty((world * (r_hi - r_lo), 2), dtype=torch.int64, device=dev)
        _all_to_all_single(rbuf.view(-1), send.contiguous().view(-1), output_split_sizes=[(r_hi - r_lo) * 2] * world,
                               input_split_sizes=[(b - a) * 2 for a, b in zip(s_lo, s_hi)], group=group)
        if world > 1:
            for src in range(world):
                recv[src * nloc + r_lo: src * nloc + r_hi] = rbuf[src * (r_hi - r_lo):(src + 1) * (r_hi - r_lo)]
    return merge_slices(recv, bounds[rank], nloc)


_UNSET = object()


def exchange_dense(agg, make_merged, device, group=None, stream=None, got=_UNSET):
    """The device wrapper of exchange_dense_tables for a DeviceAggregate whose (single) batch went through the dense path
    with an agreed code range.  Returns the merged DeviceAggregate (this rank's shard of the result) or None.
    got: the result of agg.dense_table() when the caller has already run that (final) pass."""
    if got is _UNSET:
        got = agg.dense_table(stream=stream)
    table, geo = None, (0, 0, 0, 0)
    if got is not None:
        ptr, bits, geo = got
        table = torch.as_tensor(_RawView(ptr, 2 << bits), device=device).view(-1, 2)

    def merge(recv, code0, nloc):
        world = dist.get_world_size(group)
        merged = make_merged()
        merged.merge_dense_tables(agg, [recv.data_ptr() + r * nloc * 16 for r in range(world)], code0, nloc, stream=stream)
        merged._keep = recv
        return merged
    return exchange_dense_tables(table, geo, merge, group)


# ---- small result sets: ONE all_gather, every rank merges everything ----------------------------------------------
SMALL_G_ROWS = 1 << 20   # partial groups per rank up to which the all-gather path is used


def gather_rows(rows: torch.Tensor, group=None):
    """Variable-length all_gather of row blocks [n_r, ncol]: returns (all rows grouped by source rank, counts list).
    One all_gather of the row counts + one padded all_gather of the rows."""
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    padded[: rows.shape[0]] = rows
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)]).contiguous(), counts


def exchange_allgather_small(rows: torch.Tensor, merge_rows, limit: int = SMALL_G_ROWS, group=None):
    """Small-G exchange of SURVEY.md §8(e): when every rank holds at most `limit` partial groups, the runs are
    all-gathered (a few MB) and EVERY rank merges all of them with the device merge kernel -- one collective, no
    bucketing, no all_to_all, and the merged result is replicated instead of key-sharded.  (§8e sketches the same step
    as all-gather of the keys + all_reduce of dense accumulator planes; an all_reduce would add the compensated float64
    sums (hi, lo) in the transport's own order and precision, so the partial rows travel instead and are merged by the
    library's exact merge, vnm_agg_merge_rows.)  rows: [n, n_key_words + n_acc_words] int64.
    Returns merge_rows(all_rows) or None when some rank is above the limit (all ranks agree: one all_reduce)."""
    t = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    if int(t.item()) > limit:
        return None
    allrows, _counts = gather_rows(rows, group)
    return merge_rows(allrows)


# ---- small result sets, round 5: ONE fixed-size collective ------------------------------------------------------------------
SMALL_FIXED_ROWS = 4096   # partial groups per rank the fixed-size block holds


def exchange_small_fixed(rows: torch.Tensor, merge_blocks, fixed: int = SMALL_FIXED_ROWS, group=None):
    """The small-G exchange as ONE collective (VERDICT r04 "next" #4a; exchange_allgather_small needs three -- an all_reduce to agree
    on the route, an all_gather of the row counts, a padded all_gather of the rows -- each with a host round trip: 0.57 ms of a
    4.3 ms step at G = 7).  Every rank contributes a FIXED-size block [fixed + 1, ncol]: row 0 is its header (word 0 = n, the number of
    partial groups that follow), rows 1 .. n the groups.  After the one all_gather every rank holds every header: if all counts
    fit, merge_blocks(blocks [world, fixed + 1, ncol], counts) merges them (on the device the counts are read from the headers:
    vnm_agg_merge_row_blocks -- no slicing, no concatenation), and the only host look is the world-sized copy of the counts that
    decides it.  A rank with more than `fixed` groups sends only its header; every rank sees that in the same headers and the
    function returns (None, counts): the caller takes exchange_allgather_small / the bucketed exchange, with the counts already
    known.  rows: [n, n_key_words + n_acc_words] int64.  The merged result is replicated on every rank."""
    world = dist.get_world_size(group)
    n, ncol = int(rows.shape[0]), int(rows.shape[1])
    block = torch.zeros((fixed + 1, ncol), dtype=rows.dtype, device=rows.device)
    block[0, 0] = n
    if 0 < n <= fixed:
        block[1:n + 1] = rows
    blocks = torch.empty((world, fixed + 1, ncol), dtype=rows.dtype, device=rows.device)
    if rows.is_cuda and dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(blocks.view(-1), block.view(-1), group=group)     # one flat buffer in, one out: no list of views
    else:
        _all_gather_blocks(blocks, block, group)
    counts = [int(c) for c in blocks[:, 0, 0].tolist()]          # (the one host look: world int64s)
    if max(counts) > fixed:
        return None, counts
    return merge_blocks(blocks, counts), counts


def _all_gather_blocks(blocks, block, group):
    outs = [torch.empty_like(block) for _ in range(blocks.shape[0])]
    dist.all_gather(outs, block, group=group)
    for r, o in enumerate(outs):
        blocks[r] = o


class ExchangePlan:
    """What the ranks of a standing query agreed on ONCE (agree_on_plan: one all_gather), applied to every operator the query
    creates afterwards without another collective (apply): the group-count estimate, the dense path's code range, and from the
    estimate the route of the partial-aggregate exchange -- "small" (exchange_small_fixed), "dense" (direct-addressed tables) or
    "general".  Before round 5 every step of a stream paid its agreements again (0.3 - 0.5 ms of collectives and host round trips);
    estimates only steer routes -- a wrong one costs a fallback, never a wrong result -- so the first batch's agreement serves."""

    def __init__(self, est, rng):
        self.est, self.range = int(est), rng
        self.route = "small" if 0 < self.est <= SMALL_FIXED_ROWS // 2 else ("dense" if rng is not None else "general")

    def apply(self, agg, use_estimate=True):
        if use_estimate and self.est > 0:
            agg.set_hint(self.est)
        if self.range is not None:
            agg.set_dense_range(*self.range)
        else:
            agg.set_dense_range(1, 0)


def agree_on_plan(agg, key, nrows, device, group=None, stream=None, estimate=True, hint=0) -> ExchangePlan:
    est, rng = agree_on_groups_and_range(agg, key, nrows, device, group=group, stream=stream, estimate=estimate)
    return ExchangePlan(est if estimate else hint, rng)


# ---- ORDER BY ... LIMIT K over batch-sharded rows --------------------------------------------------------------------
def topk_exchange(values: torch.Tensor, row_ids: torch.Tensor, k: int, descending: bool, group=None):
    """Distributed top-K: every rank passes its LOCAL first-K rows of `ORDER BY v [DESC] LIMIT k`
    (values + GLOBAL row ids, e.g. from vnm_sort_indices with limit = k over its shard); one all_gather of K rows per
    rank, then the same ordering rule selects the global first K on every rank: values first, NaN last in both
    directions, ties in ascending global row id (the order Sort::Sorted gives over the concatenated table,
    sort.cpp:22-37 -- stable).  Returns (values, row_ids) of the k winners, identical on all ranks."""
    rows = torch.stack([values.view(torch.int64) if values.dtype == torch.float64 else values.to(torch.int64), row_ids.to(torch.int64)], dim=1)
    allrows, _ = gather_rows(rows.contiguous(), group)
    v = allrows[:, 0].view(torch.float64) if values.dtype == torch.float64 else allrows[:, 0]
    ids = allrows[:, 1]
    order = torch.argsort(ids, stable=True)                                    # ties: ascending global row id
    v, ids = v[order], ids[order]
    if v.dtype == torch.float64:
        nan = torch.isnan(v)
        key = torch.where(nan, torch.zeros_like(v), v)
        key = torch.where(key == 0, torch.zeros_like(key), key)                # -0.0 == +0.0 (Arrow's comparison)
        order = torch.argsort(key, descending=descending, stable=True)
        v, ids, nan = v[order], ids[order], nan[order]
        order = torch.argsort(nan.to(torch.int8), stable=True)                 # NaN after every number, both directions
        v, ids = v[order], ids[order]
    else:
        order = torch.argsort(v, descending=descending, stable=True)
        v, ids = v[order], ids[order]
    return v[:k], ids[:k]


# ---- ORDER BY over batch-sharded rows: distributed sample sort -----------------------------------------------------------
def _order_key(v: torch.Tensor, descending: bool) -> torch.Tensor:
    """A float64 / int64 key as an int64 whose ASCENDING order is the requested order with NaN last in both directions and
    -0.0 == +0.0 (Arrow's SortIndices, sort.cpp:22-37): the order-preserving code of the single-GPU sort (vnm_sort.hip:
    encode_key), computed with torch ops so that the same function serves the CPU (gloo) tests."""
    if v.dtype == torch.float64:
        x = torch.where(v == 0, torch.zeros_like(v), v)               # -0.0 -> +0.0
        bits = x.view(torch.int64)
        code = torch.where(bits < 0, ~bits, bits | torch.iinfo(torch.int64).min)    # unsigned order-preserving image ...
        code = code ^ torch.iinfo(torch.int64).min                                   # ... as a signed int64 with the same order
        if descending:
            code = ~code
        return torch.where(torch.isnan(v), torch.full_like(code, torch.iinfo(torch.int64).max), code)
    code = v.to(torch.int64)
    return ~code if descending else code


def ssort_sample(code: torch.Tensor, m: int) -> torch.Tensor:
    """strided sample of at most m order codes"""
    n = int(code.shape[0])
    m = min(n, m)
    if not m:
        return code[:0]
    return code[(torch.arange(m, device=code.device, dtype=torch.int64) * n) // m]


def ssort_splitters(samples, world: int) -> torch.Tensor:
    """world - 1 splitters: the sorted union of the ranks' samples cut into `world` equal parts"""
    union, _ = torch.sort(torch.cat(list(samples)))
    if not len(union) or world <= 1:
        return union[:0]
    cut = (torch.arange(1, world, device=union.device, dtype=torch.int64) * len(union)) // world
    return union[cut]


def ssort_owner(code: torch.Tensor, splitters: torch.Tensor) -> torch.Tensor:
    """rank whose splitter range holds the code (code == splitter -> the upper rank: equal codes never split)"""
    if not len(splitters):
        return torch.zeros_like(code)
    return torch.searchsorted(splitters, code, right=True)


def partition_by_owner(code: torch.Tensor, splitters: torch.Tensor, world: int):
    """Stable partition of a rank's rows by owner: (row numbers with owner 0's rows first and the source order kept inside an owner,
    rows per owner).  On the device the library's kernels (vnm_partition_by_owner: per-block owner histograms, a scan per owner, a
    stable scatter of the row numbers -- no sort); on the host (the gloo tests) a stable argsort of the owners."""
    n = int(code.shape[0])
    if code.is_cuda and world <= 64:
        import ctypes
        from . import _lib as L
        code = code.contiguous()
        sp = splitters.to(torch.int64).contiguous()
        order = torch.empty(max(n, 1), dtype=torch.int64, device=code.device)
        counts = torch.zeros(world, dtype=torch.int64, device=code.device)
        stream = torch.cuda.current_stream(code.device).cuda_stream
        L.check(L.lib().vnm_partition_by_owner(ctypes.c_void_p(code.data_ptr()), n, ctypes.c_void_p(sp.data_ptr() if len(sp) else 0), int(len(sp)),
                                               ctypes.c_void_p(order.data_ptr()), ctypes.c_void_p(counts.data_ptr()), ctypes.c_void_p(stream)))
        return order[:n], counts
    owner = ssort_owner(code, splitters)
    return torch.argsort(owner, stable=True), torch.bincount(owner, minlength=world)


def sample_sort_exchange(keys: torch.Tensor, global_ids: torch.Tensor, descending: bool, sort_local, group=None,
                         samples_per_rank: int = 4096):
    """Distributed `ORDER BY key [DESC]` (Sort::Sorted semantics, sort.cpp:15-63: stable, NaN after every
    number in both directions) over rows sharded by batch: rank r holds keys[r] with their GLOBAL row ids (ranks hold
    contiguous, ascending id ranges).
      1. every rank contributes a strided sample of its order codes; one all_gather; the sorted union is cut into `world`
         equal parts: world - 1 SPLITTERS, the same on every rank;
      2. rows go to the rank whose splitter range holds their code (rows of equal code always meet on one rank): one
         variable-size all_to_all of (code, global id, key bits) triples, blocks in source-rank order;
      3. sort_local(codes, ids) -> permutation: the receiving rank sorts what it got by (code, id) -- on the GPU the library's
         single-key sort over the codes (stable over blocks that arrive in id order), in the CPU tests a torch sort.
    Returns (keys, global_ids) of this rank's slice of the global order: concatenating the ranks' results in rank order IS the
    sorted table (rank 0 holds the first rows)."""
    world = dist.get_world_size(group)
    code = _order_key(keys, descending)
    samp = ssort_sample(code, samples_per_rank)
    m = int(samp.shape[0])
    padded = torch.full((samples_per_rank + 1,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=code.device)
    padded[:m] = samp
    padded[samples_per_rank] = m
    allp = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(allp, padded, group=group)
    splitters = ssort_splitters([p_[: int(p_[samples_per_rank])] for p_ in allp], world)
    order, counts = partition_by_owner(code, splitters, world)           # by owner, source row order kept
    send = torch.stack([code[order], global_ids[order].to(torch.int64),
                        (keys.view(torch.int64) if keys.dtype == torch.float64 else keys.to(torch.int64))[order]], dim=1).contiguous()
    recv = exchange(send, counts, group)
    rc, rid, rk = recv[:, 0].contiguous(), recv[:, 1].contiguous(), recv[:, 2].contiguous()
    perm = sort_local(rc, rid)
    out_keys = rk[perm]
    return (out_keys.view(torch.float64) if keys.dtype == torch.float64 else out_keys.to(keys.dtype)), rid[perm]


# ---- string (non-numeric) group keys across ranks (round 4; SURVEY.md 8 f3 x 8e) -------------------------------------------------
# GenericHashAggregate keys its map on the VALUES (generic_hash_aggregate.h:10-45).  Here every rank dictionary-encodes a
# non-numeric key column on its own (vnm_strdict_encode): code 7 is "Berlin" on one rank and "Riva" on another, so partial groups
# cannot be exchanged by code.  Before the exchange the ranks build ONE dictionary: every rank contributes its values (one
# all_gather of the Arrow arrays' buffers), the union keeps them in (rank, local order) of first appearance -- the same on every
# rank -- and each rank re-keys its partial groups by union id (one gather through a remap table).  From then on the key word is
# an ordinary integer: owner_of / bucket_by_owner / the exchanges above apply unchanged, and the result's key column is decoded
# through the union dictionary.

def union_dictionary(values, group=None):
    """values: this rank's dictionary as a pyarrow array (distinct values, position = local code).
    Returns (union: pa.Array of the distinct values of all ranks, remap: np.int32 array with remap[local code] = union id).
    Deterministic and identical on every rank: rank 0's values in its order, then the values rank 1 adds, ..."""
    import numpy as np
    import pyarrow as pa
    import pyarrow.compute as pc
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sink = pa.BufferOutputStream()
    t = pa.table({"v": values})
    with pa.ipc.new_stream(sink, t.schema) as w:
        w.write_table(t)
    blob = sink.getvalue().to_pybytes()
    blobs = [None] * world
    dist.all_gather_object(blobs, blob, group=group)
    parts = [pa.ipc.open_stream(b).read_all().column(0).combine_chunks() for b in blobs]
    return union_of_dictionaries(parts, rank)


def union_of_dictionaries(parts, rank):
    """The pure part of union_dictionary: parts[r] = rank r's dictionary (position = local code; a NULL entry = a code that was
    never handed out -> remap -1).  Returns (union, remap of `rank`)."""
    import numpy as np
    import pyarrow as pa
    import pyarrow.compute as pc
    # ONE type for every rank's part: a rank without rows brings a null-typed (or differently typed) empty dictionary, string and
    # large_string dictionaries may meet -- the first part with a real type decides (a large variant anywhere wins over the small one)
    typed = [p.type for p in parts if not pa.types.is_null(p.type)]
    ut = typed[0] if typed else pa.null()
    for t in typed:
        if pa.types.is_large_string(t) or pa.types.is_large_binary(t):
            ut = t
            break
    parts = [p if p.type == ut else (pa.nulls(len(p), ut) if pa.types.is_null(p.type) else p.cast(ut)) for p in parts]
    union = pa.array([], type=ut)
    remap = None
    for r, part in enumerate(parts):
        valid = part.is_valid().to_numpy(zero_copy_only=False) if len(part) else np.zeros(0, bool)
        pos = np.full(len(part), -1, np.int64)
        if valid.any():
            live = part.filter(pa.array(valid))
            idx = pc.index_in(live, value_set=union) if len(union) else pa.nulls(len(live), pa.int32())
            known = idx.is_valid().to_numpy(zero_copy_only=False)
            p_live = idx.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int64)
            if not known.all():
                # the values new to the union, each ONCE (a dictionary that repeats a value would otherwise split its group): the
                # position of a new value = the union's length + its rank among the distinct new values, in first-appearance order
                fresh = live.filter(pa.array(~known))
                distinct = pc.unique(fresh)
                p_live[~known] = len(union) + pc.index_in(fresh, value_set=distinct).to_numpy(zero_copy_only=False).astype(np.int64)
                union = pa.concat_arrays([union, distinct])
            pos[valid] = p_live
        if r == rank:
            remap = pos.astype(np.int32)
    if len(union) >= 2**31:
        raise RuntimeError("union_dictionary: more than 2^31 distinct key values")
    return union, remap


def rekey_codes(codes: torch.Tensor, null_mask: torch.Tensor, remap) -> torch.Tensor:
    """Key word of the partial groups (local dictionary codes; the NULL group's word is 0 with its mask set) -> union ids.
    codes / null_mask: int64 tensors (host or device); remap: what union_dictionary returned."""
    table = torch.as_tensor(remap, dtype=torch.int64, device=codes.device)
    if table.numel() == 0:
        return codes.clone()
    safe = torch.where(null_mask != 0, torch.zeros_like(codes), codes)
    out = table[safe.clamp_(0, table.numel() - 1)]
    return torch.where(null_mask != 0, torch.zeros_like(out), out)
