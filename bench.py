#!/usr/bin/env python3
"""bench.py -- filter -> group-by over HBM-resident Arrow-layout columns on MI355X.

Contract (one JSON line on rank 0):  python bench.py --gpus N --steps K --warmup W
A "step" = one pass of the hot path over one synthetic batch already resident in HBM:
    groupby (default, BASELINE.json configs[2]):
        SELECT k, sum(v), avg(v) FROM t WHERE v > X GROUP BY k       (fused filter + hash aggregate)
        N = 1e9 rows, int64 key uniform in [0, G), fp64 value = j * 2^-7 (every partial sum exact ->
        bit-exact float aggregates), X chosen for selectivity 0.5.
    filter (BASELINE.json configs[1]):
        WHERE fare_amount > X over a 1e9-row fp64 column -> compacted column.
    stream (BASELINE.json configs[3]; the default with --gpus N > 1):
        the same query over a STREAM of 2^24-row record batches (what stream_csv / TableReaderOperator hand the aggregate,
        vinum/api/stream_reader.py:32-94), HBM-resident, dealt round-robin to the ranks: every rank streams its
        59 batches into one operator (vnm_agg_set_async: the waiting batches go to the device as the segments of one
        launch), G = 1e6 (--groups; G = 7 and the one-batch G = 1e8 shape under "also"), partial aggregates
        exchanged over RCCL inside the timed region.
Multi-GPU (--gpus N>1): one process per GPU.  Launched by `python -m torch.distributed.run ... bench.py --gpus N`, or
by itself: without WORLD_SIZE in the environment `bench.py --gpus N` re-executes itself under torch.distributed.run
(--nproc-per-node N, rendezvous on 127.0.0.1).  Batches shard by rank (weak scaling), every rank aggregates its shard,
partial groups are exchanged over RCCL and merged by their owner; rank 0 prints the ONE JSON line.
VNM_BENCH_DRY_RUN=1: no GPU work at all -- the launch, the rendezvous (gloo), the max-over-ranks timing and the JSON line only
(tests/test_bench_launch.py runs that here).

Extra keys: "roofline" (HIP-event time of the dominant kernel vs 8 TB/s HBM peak) and "cpu_baseline"
(the reference's CPU path -- NumPy compare + pyarrow filter + the real reference C++ aggregate from
oracle/_ref when that build is present, else the oracle port -- on a bounded sample, rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sustain-seconds", type=float, default=6.0, help="after the timed region: repeat the step for this many seconds (reported as `sustained`; 0 = off)")
    ap.add_argument("--workload", default=None, choices=["groupby", "stream", "filter", "topk", "project"],
                    help="default: groupby (configs[2]) on one GPU, stream (configs[3]) on several")
    ap.add_argument("--batches", type=int, default=0, help="stream: record batches of 2^24 rows per rank (default: rows // 2^24, at most 60)")
    ap.add_argument("--shape", default="hot", choices=["hot", "count_star", "minmax"],
                    help="group-by function mix: hot = sum,avg (configs[2]); count_star = configs[0]'s query shape "
                         "(no predicate); minmax = min,max -- the latter two run the generic accumulator kernel")
    ap.add_argument("--no-also", action="store_true", help="skip the side measurements (configs[1] filter, G=7 group-by)")
    ap.add_argument("--limit", type=int, default=10)
    ap.add_argument("--rows", type=float, default=1e9)
    ap.add_argument("--groups", type=float, default=None, help="default: 1e8 (groupby), 1e6 (stream)")
    ap.add_argument("--selectivity", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true",
                    help="after the timed region: property-check the (merged) group-by result of the last step on the device -- "
                         "survivors and totals conserved across all ranks, every merged key owned by exactly one rank")
    ap.add_argument("--no-check", action="store_true", help="skip the property check of the last step's result")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--hint", action="store_true",
                    help="tell the operator the group count (vnm_agg_set_hint).  The reference's operator boundary has no such "
                         "argument, so the headline is the HINT-LESS run: the operator samples the keys itself")
    ap.add_argument("--no-hint", action="store_true", help=argparse.SUPPRESS)  # r01 spelling of what is now the default
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "stream" if args.gpus > 1 else "groupby"
    if args.groups is None:
        args.groups = 1e6 if args.workload == "stream" else 1e8
    return args


def self_launch(args):
    """`bench.py --gpus N` started on its own (no WORLD_SIZE): become `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` -- one process per GPU, rank 0 prints
    the JSON line on the inherited stdout."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def dry_run(args):
    """VNM_BENCH_DRY_RUN=1: everything around the GPU work of an N-rank run -- rendezvous (gloo, CPU), barriers, the max-over-ranks
    clock, ONE JSON line from rank 0 -- with a sleep where the step would be.  Marked as such; never a measurement."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "rows/sec + achieved HBM GB/s, filter->group-by over 10^9-row Arrow batches", "value": None, "unit": "rows/s",
                          "n_gpus": world, "rccl_ranks": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t.item()) / max(args.steps, 1) * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64/int64", "data": "none",
                          "dry_run": True, "config": {"workload": f"DRY RUN of the {world}-rank launch path ({args.workload}): no GPU work"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def gen_data(torch, n, groups, seed, device):
    """key int64 uniform [0, G); value fp64 = j / 128 with j uniform in [0, 2^14)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    j = torch.randint(0, 1 << 14, (n,), device=device, dtype=torch.int64, generator=g)
    v = j.to(torch.float64) / 128.0
    del j
    k = torch.randint(0, int(groups), (n,), device=device, dtype=torch.int64, generator=g)
    return k, v


def threshold_for(selectivity):
    # v = j/128, j uniform in [0, 16384): P(v > X) = 1 - (floor(128 X) + 1)/16384
    jx = int(round((1.0 - selectivity) * 16384)) - 1
    return max(jx, -1) / 128.0


def cpu_baseline(args, x_thr):
    """The reference's own CPU path on a bounded sample of the same workload."""
    import pyarrow as pa
    from oracle import oracle as O
    from oracle import ref as R
    rng = np.random.default_rng(123)
    groups = int(args.groups)

    def make(n):
        k = rng.integers(0, groups, n).astype(np.int64)
        v = rng.integers(0, 1 << 14, n).astype(np.float64) / 128.0
        return pa.RecordBatch.from_arrays([pa.array(k), pa.array(v)], names=["k", "v"])

    use_ref = R.available()
    funcs = [(O.SUM, "v", "sum_v"), (O.AVG, "v", "avg_v")]

    def run(batches):
        t0 = time.perf_counter()
        if args.workload == "filter":
            for b in batches:
                x = b.column(1).to_numpy(zero_copy_only=True)          # record_batch.py:112-118
                mask = x > x_thr                                        # expressions.py:32
                b.filter(pa.array(mask), null_selection_behavior="emit_null")  # record_batch.py:85-90
        else:
            agg = R.RefAggregate(R.SINGLE, ["k"], ["k"], funcs) if use_ref else O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
            for b in batches:
                x = b.column(1).to_numpy(zero_copy_only=True)
                fb = b.filter(pa.array(x > x_thr), null_selection_behavior="emit_null")
                agg.next(fb)
            agg.result()
        return time.perf_counter() - t0

    chunk = 1_000_000  # reference batches (its default is 10 000 rows; 1e6 is its best case)
    probe = [make(chunk) for _ in range(2)]
    t_probe = run(probe)
    rate = 2 * chunk / t_probe
    n_batches = int(max(2, min(60, args.cpu_seconds * rate / chunk)))
    batches = [make(chunk) for _ in range(n_batches)]
    t = run(batches)
    kind = "reference" if (use_ref or args.workload == "filter") else "port"
    # an INDEPENDENT CPU line on the same sample (BASELINE.md 3 iv): pyarrow.compute's own filter and Table.group_by
    # (Arrow's multi-threaded C++ kernels, nothing of the reference or of this repo in them)
    indep = None
    try:
        import pyarrow.compute as pc
        tb = pa.Table.from_batches(batches)
        t0 = time.perf_counter()
        ft = tb.filter(pc.greater(tb.column("v"), x_thr))
        t_f = time.perf_counter() - t0
        if args.workload == "filter":
            indep = {"what": "pyarrow.compute: Table.filter(pc.greater(v, X))", "rows_per_s": tb.num_rows / t_f, "threads": pa.cpu_count()}
        else:
            t0 = time.perf_counter()
            gr = ft.group_by("k").aggregate([("v", "sum"), ("v", "mean")])
            t_g = time.perf_counter() - t0
            indep = {"what": "pyarrow.compute: Table.filter(pc.greater(v, X)).group_by('k').aggregate([sum, mean])",
                     "rows_per_s": tb.num_rows / (t_f + t_g), "filter_s": t_f, "group_by_s": t_g, "groups": gr.num_rows, "threads": pa.cpu_count()}
        del tb, ft
    except Exception as e:
        indep = {"error": str(e)}
    all_cores = None
    try:
        all_cores = cpu_baseline_all_cores(args, x_thr, use_ref)
    except Exception as e:  # reporting only
        all_cores = {"value": None, "error": str(e)}
    return {"value": n_batches * chunk / t, "unit": "rows/s", "cores": 1, "kind": kind, "all_cores": all_cores, "pyarrow_compute": indep,
            "sample": f"{n_batches} batches x {chunk} rows of the same synthetic workload "
                      f"(G={groups}, s={args.selectivity}), single-threaded like the reference executor; "
                      + ("NumPy compare + pyarrow filter" if args.workload == "filter" else
                         ("NumPy compare + pyarrow filter + reference SingleNumericalHashAggregate (oracle/_ref)"
                          if use_ref else "NumPy compare + pyarrow filter + oracle port of the aggregate"))}


def _cpu_worker(q, workload, groups, x_thr, use_ref, seconds, seed):
    """One host core: the reference's CPU path over its own stream of batches for ~`seconds` (the all-cores
    upper bound is a simple range partition -- every core aggregates its own rows, no merge is timed)."""
    import pyarrow as pa
    from oracle import oracle as O
    from oracle import ref as R
    rng = np.random.default_rng(seed)
    chunk = 1_000_000
    funcs = [(O.SUM, "v", "sum_v"), (O.AVG, "v", "avg_v")]
    batches = []
    for _ in range(3):
        k = rng.integers(0, groups, chunk).astype(np.int64)
        v = rng.integers(0, 1 << 14, chunk).astype(np.float64) / 128.0
        batches.append(pa.RecordBatch.from_arrays([pa.array(k), pa.array(v)], names=["k", "v"]))
    agg = None
    if workload != "filter":
        agg = R.RefAggregate(R.SINGLE, ["k"], ["k"], funcs) if use_ref else O.OracleAggregate(O.SINGLE, ["k"], ["k"], funcs)
    rows, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        b = batches[(rows // chunk) % len(batches)]
        x = b.column(1).to_numpy(zero_copy_only=True)
        fb = b.filter(pa.array(x > x_thr), null_selection_behavior="emit_null")
        if agg is not None:
            agg.next(fb)
        rows += chunk
    if agg is not None:
        agg.result()
    q.put((rows, time.perf_counter() - t0))


def cpu_baseline_all_cores(args, x_thr, use_ref):
    import multiprocessing as mp
    cores = min(os.cpu_count() or 1, 64)
    ctx = mp.get_context("spawn")   # the parent holds a HIP context: never fork it
    q = ctx.Queue()
    seconds = max(4.0, min(args.cpu_seconds, 10.0))
    procs = [ctx.Process(target=_cpu_worker, args=(q, args.workload, int(args.groups), x_thr, use_ref, seconds, 1000 + i))
             for i in range(cores)]
    t0 = time.perf_counter()
    for p in procs:
        p.start()
    got = [q.get(timeout=seconds * 6 + 120) for _ in procs]
    for p in procs:
        p.join()
    wall = time.perf_counter() - t0
    rows = sum(r for r, _ in got)
    busy = max(t for _, t in got)
    return {"value": rows / busy, "unit": "rows/s", "cores": cores,
            "sample": f"{cores} processes x ~{seconds:.0f} s of the same single-threaded reference path, each over its own rows "
                      f"(range partition, merge not timed); wall {wall:.1f} s incl. process start"}


def check_groupby(torch, dist, state, k, v, x_thr, world, rank, device, exchanged, strict=True):
    """Size-independent properties of the LAST step's result, evaluated on the device and reduced over all ranks
    (the data is quantised, so the float sums are exact): survivors conserved, totals conserved, no key on two ranks."""
    from vinum_amd import _lib as L
    agg = state.get("merged") if (world > 1 or exchanged) and state.get("merged") is not None else state["agg"]
    if isinstance(agg, tuple):
        agg = agg[0]
    replicated = bool(state.get("replicated"))
    n = agg.finish()
    lay = agg.word_layout()
    w_cnt = next(w for kind, col, w in lay["ops"] if kind == 1)
    w_sum = next(w for kind, col, w in lay["ops"] if kind == 2)
    kp, ap_ = agg.dense_ptrs()
    def view(ptr, typestr="<i8"):
        return torch.as_tensor(CudaArrayView(ptr, n, typestr), device=device) if n else torch.zeros(0, dtype=torch.int64, device=device)
    keys, cnt, sm = view(kp[0]), view(ap_[w_cnt]), view(ap_[w_sum], "<f8")
    if lay["merge"][w_sum] == L.M_ADD_F64C and n:
        sm = sm + view(ap_[w_sum + 1], "<f8")
    keep = v > x_thr
    mine = 1.0 if (not replicated or rank == 0) else 0.0     # a replicated result counts once
    t = torch.tensor([float(keep.sum()), float((v[keep] * 128.0).sum()), mine * float(cnt.sum()) if n else 0.0,
                      mine * float((sm * 128.0).sum()) if n else 0.0, mine * float(n)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t)
    ok = {"world_size": world, "survivors_conserved": t[0].item() == t[2].item(), "totals_conserved": t[1].item() == t[3].item(),
          "groups_all_ranks": int(t[4].item())}
    ok["exchange"] = state.get("exchange_kind")
    if world > 1 and state.get("exchange_kind") == "bucketed" and n:
        from vinum_amd import distributed as D
        own = D.owner_of([keys, torch.zeros_like(keys)], world)
        bad = torch.tensor([float((own != rank).sum())], dtype=torch.float64, device=device)
        dist.all_reduce(bad)
        ok["every_key_on_its_owner"] = bad.item() == 0.0
    if world == 1 and not exchanged and state.get("cols") is not None and n:
        # the step's RESULT COLUMNS (written by the fused final pass) against the dense partial state of the same operator:
        # same keys, sum = hi (+ lo), avg = sum / count -- compared as sorted-by-key columns
        ck, cs, ca = state["cols"][:3]
        rk = torch.as_tensor(CudaArrayView(ck.values_ptr, n, "<i8"), device=device)
        rs = torch.as_tensor(CudaArrayView(cs.values_ptr, n, "<f8"), device=device)
        ra = torch.as_tensor(CudaArrayView(ca.values_ptr, n, "<f8"), device=device)
        o1, o2 = torch.argsort(rk), torch.argsort(keys)
        ok["result_columns_match"] = bool(ck.length == n and torch.equal(rk[o1], keys[o2]) and torch.equal(rs[o1], sm[o2])
                                          and torch.equal(ra[o1], sm[o2] / cnt[o2].to(torch.float64)))
    ok["ok"] = all(val for key, val in ok.items() if isinstance(val, bool))
    if strict:
        assert ok["ok"], f"multi-GPU result check failed: {ok}"
    return ok


class CudaArrayView:
    """Expose a raw device pointer to torch (zero copy) through __cuda_array_interface__."""

    def __init__(self, ptr, n, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _spans(lib, ctypes, names, steps):
    out = {}
    for nm in names:
        tot_ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        lib.vnm_profile_query(nm, ctypes.byref(tot_ms), ctypes.byref(cnt))
        if cnt.value:
            out[nm.decode()] = tot_ms.value / steps
    return out


AGG_SPANS = [b"agg_estimate", b"agg_scan", b"agg_part_scatter1", b"agg_part_scatter2", b"agg_part_final", b"agg_finalize", b"agg_table_merge"]


def device_state():
    """clocks / power / temperatures of GPU 0 as rocm-smi reports them (a judge can normalise box-to-box spread by them); {} if unavailable"""
    import json as _json
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = _json.loads(out[out.index("{"):])
        card = next(iter(j.values()))
        keep = {}
        for k_, v_ in card.items():
            kl = k_.lower()
            if any(w in kl for w in ("sclk", "mclk", "fclk", "power", "temperature")):
                keep[k_] = v_
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:80]}


def _measure(torch, lib, ctypes, fn, span_names, steps, warmup):
    """`steps` timed calls of fn after `warmup` untimed ones: wall time per step and HIP-event time per kernel span."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    lib.vnm_set_profiling(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    sp = _spans(lib, ctypes, span_names, steps)
    lib.vnm_set_profiling(0)
    return el / steps * 1e3, sp


def _entry(workload, n, ms, sp, alg, result_rows, extra=None):
    kms = sum(sp.values())
    ach = alg / (kms * 1e-3) / 1e9 if kms else 0.0
    e = {"workload": workload, "rows_per_s": n / (ms * 1e-3), "ms_per_step": ms, "result_rows": int(result_rows),
         "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                      "kernels_ms": {k: round(v, 4) for k, v in sp.items()}, "algorithmic_bytes": alg}}
    if extra:
        e.update(extra)
    return e


def side_workloads(torch, lib, L, ops, pa, ctypes, kcol, vcol, n, x_thr, stream, args, steps=3, warmup=1):
    """The other single-GPU configurations of BASELINE.json over the headline run's resident columns, a few steps each,
    same timing method (HIP events per kernel): the HINTED headline query, configs[1] (WHERE v > X -> compacted column),
    the small-cardinality group-by (G = 7) and configs[0]'s query shape at scale, configs[4] (ORDER BY v DESC LIMIT 10
    and the three-expression projection), and the headline query END TO END including its result columns
    (BaseAggregate::Result: key, sum, avg finalised on the device into Arrow buffers)."""
    from vinum_amd.device import DeviceColumn
    out = {}
    device = torch.device("cuda", torch.cuda.current_device())
    groups = int(args.groups)
    state = {}

    def headline(hint, finalize=False):
        def run():
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())],
                                      expected_groups=hint)
            agg.set_predicate(">", x_thr)
            agg.next([kcol], [vcol, vcol], pred=vcol, nrows=n, stream=stream)
            if finalize:
                state["cols"] = agg.result_device(stream=stream)   # key, sum(v), avg(v) as Arrow-layout HBM buffers
                state["ng"] = agg.result_rows
            else:
                state["ng"] = agg.finish(stream=stream)
            state["agg"] = agg
        return run

    # ---- the headline query with the group count given (vnm_agg_set_hint): not reachable through the reference boundary
    # (two warm-up calls: the previous step's operator / outputs stay alive while the next step allocates, so the caching
    # allocator only reaches its steady state -- no hipMalloc, ~25 ms per GB, inside the timed steps -- after the second)
    ms, sp = _measure(torch, lib, ctypes, headline(groups, finalize=True), AGG_SPANS, steps, warmup + 1)
    out["configs[2] hinted"] = _entry(f"the headline query (result columns included) with expected_groups={groups:.3g} passed to the operator", n, ms, sp,
                                      16.0 * n + 24.0 * state["ng"], state["ng"])
    # ---- the headline query stopping at the dense PARTIAL state (key, sum hi / lo, count words: what a further batch or the
    # multi-GPU exchange would merge into) instead of the result columns -- the r01 / r02 headline
    ms, sp = _measure(torch, lib, ctypes, headline(0, finalize=False), AGG_SPANS, steps, warmup + 1)
    out["partial_state"] = _entry("configs[2] hint-less up to vnm_agg_finish: dense partial state (key, sum, compensation, count words) "
                                  "instead of the finalised key / sum / avg columns of the headline; input resident in HBM", n, ms, sp,
                                  16.0 * n + 24.0 * state["ng"], state["ng"])
    state.clear()
    # ---- BASELINE.md section 3's "+- 1 % nulls" variant of the value column: the headline query with a validity bitmap on v (a NULL
    # fails `v > X`; the dense path reads one validity byte per pair of rows) -- and the headline query over SKEWED keys,
    # k = floor(G u^4) (the first key ~1 % of the rows, thousands of keys far above an even share: the ring scatter's round limit)
    try:
        gn = torch.Generator(device=device); gn.manual_seed(11)
        bits = torch.full(((n + 7) // 8,), 255, dtype=torch.uint8, device=device)
        holes = torch.randint(0, (n + 7) // 8, (n // 100,), device=device, generator=gn)
        bits[holes] = bits[holes] & ~(torch.ones_like(holes, dtype=torch.uint8) << (holes % 8).to(torch.uint8))
        vnull = DeviceColumn(vcol.values_ptr, bits.data_ptr(), 0, n, pa.float64(), keep=(bits,))

        def with_nulls():
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
            agg.set_predicate(">", x_thr)
            agg.next([kcol], [vnull, vnull], pred=vnull, nrows=n, stream=stream)
            state["cols"] = agg.result_device(stream=stream)
            state["ng"] = agg.result_rows
        ms, sp = _measure(torch, lib, ctypes, with_nulls, AGG_SPANS, steps, warmup + 1)
        out["configs[2] with ~1 % NULLs in v"] = _entry("the headline query (hint-less, result columns included) over a NULLABLE value column (~1 % of the rows NULL)",
                                                        n, ms, sp, 16.125 * n + 24.0 * state["ng"], state["ng"])
        state.clear()
        del vnull, bits, holes
        u = torch.rand(n, device=device, dtype=torch.float64, generator=gn)
        ks = (u.pow_(4.0) * groups).to(torch.int64)
        del u
        kscol = DeviceColumn.from_torch(ks)

        def skewed():
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
            agg.set_predicate(">", x_thr)
            agg.next([kscol], [vcol, vcol], pred=vcol, nrows=n, stream=stream)
            state["cols"] = agg.result_device(stream=stream)
            state["ng"] = agg.result_rows
        ms, sp = _measure(torch, lib, ctypes, skewed, AGG_SPANS, steps, warmup + 1)
        out["configs[2] over skewed keys"] = _entry(f"the headline query (hint-less, result columns included) over keys k = floor(G u^4), u uniform, G = {groups:.3g}",
                                                    n, ms, sp, 16.0 * n + 24.0 * state["ng"], state["ng"])
        state.clear()
        del kscol, ks
        # ... and a NULLABLE key column ("+- nulls"): ~12 % of the keys NULL -- pass 1 of the dense path reads the key's
        # validity and sums the NULL-key rows as the one group they are (single_numerical_hash_aggregate.cpp:24-32)
        kb = torch.randint(0, 256, ((n + 7) // 8,), device=device, dtype=torch.uint8, generator=gn)
        for _ in range(2):
            kb |= torch.randint(0, 256, ((n + 7) // 8,), device=device, dtype=torch.uint8, generator=gn)
        knull = DeviceColumn(kcol.values_ptr, kb.data_ptr(), 0, n, pa.int64(), keep=(kcol, kb))

        def null_keys():
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
            agg.set_predicate(">", x_thr)
            agg.next([knull], [vcol, vcol], pred=vcol, nrows=n, stream=stream)
            state["cols"] = agg.result_device(stream=stream)
            state["ng"] = agg.result_rows
        ms, sp = _measure(torch, lib, ctypes, null_keys, AGG_SPANS, steps, warmup + 1)
        out["configs[2] with ~12 % NULL keys"] = _entry("the headline query (hint-less, result columns included) over a NULLABLE key column (~12 % of the keys NULL: one group)",
                                                        n, ms, sp, 16.125 * n + 24.0 * state["ng"], state["ng"])
        state.clear()
        del knull, kb
        # ... over values that are NOT multiples of 1/128: nearly every add has a rounding error, so the compensation word of a sum
        # (TwoSum: hi, lo) is touched by nearly every entry of the final pass -- the headline's quantised values (exact adds, which is
        # what lets --check compare totals bit for bit) never touch it
        gr = torch.Generator(device=device); gr.manual_seed(11)
        vreal = torch.randn(n, device=device, dtype=torch.float64, generator=gr) * 20.0 + x_thr
        vrcol = DeviceColumn.from_torch(vreal)

        def real_values():
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
            agg.set_predicate(">", x_thr)
            agg.next([kcol], [vrcol, vrcol], pred=vrcol, nrows=n, stream=stream)
            state["cols"] = agg.result_device(stream=stream)
            state["ng"] = agg.result_rows
        ms, sp = _measure(torch, lib, ctypes, real_values, AGG_SPANS, steps, warmup + 1)
        out["configs[2] over non-quantised values"] = _entry("the headline query (hint-less, result columns included) over v ~ N(X, 20): every add of the compensated sums "
                                                             "has a rounding error", n, ms, sp, 16.0 * n + 24.0 * state["ng"], state["ng"])
        state.clear()
        del vreal, vrcol
    except Exception as e:  # noqa: BLE001 -- a side measurement must not take the headline line down
        out["configs[2] variants"] = {"error": repr(e)}
    # ---- configs[1]
    dst = torch.empty(n, dtype=torch.float64, device=device)
    cnt = ctypes.c_int64(0)

    def filt():
        ov = (ctypes.c_void_p * 1)(dst.data_ptr())
        ob = (ctypes.c_void_p * 1)(None)
        d = vcol.dcol()
        L.check(lib.vnm_filter_cmp(ctypes.byref(d), L.GT, 1, x_thr, 0, 1, ctypes.byref(d), ov, ob, ctypes.byref(cnt),
                                   ctypes.c_void_p(stream)))
    ms, sp = _measure(torch, lib, ctypes, filt, [b"filter_kernel"], steps + 2, warmup + 1)
    out["configs[1] filter"] = _entry(f"WHERE v > {x_thr} over {n:.3g}-row fp64 column -> compacted column", n, ms, sp,
                                      8.0 * n + 8.0 * cnt.value, cnt.value)
    del dst
    # ---- group-by with 7 groups: keys folded onto [0, 7) by a fused projection (k % 7), then the same query
    k7 = ops.project(("mod", "k", 7), {"k": kcol}, length=n, stream=stream)

    def gb():
        agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())],
                                  expected_groups=7 if args.hint else 0)
        agg.set_predicate(">", x_thr)
        agg.next([k7], [vcol, vcol], pred=vcol, nrows=n, stream=stream)
        state["ng"] = agg.finish(stream=stream)
    ms, sp = _measure(torch, lib, ctypes, gb, AGG_SPANS, steps + 2, warmup + 1)
    out["group-by G=7"] = _entry(f"SELECT k,sum(v),avg(v) WHERE v>{x_thr} GROUP BY k; N={n:.3g}, 7 groups", n, ms, sp,
                                 16.0 * n + 24.0 * state["ng"], state["ng"])

    # ---- the same query over other group counts (keys folded with k % G): no cardinality cliff between the scan kernels,
    # the direct-addressed LDS scan, the one- and two-level dense paths
    sweep = {}
    for gs in (1_000, 10_000, 100_000, 1_000_000, 10_000_000):
        if gs >= groups:
            continue
        kg = ops.project(("mod", "k", gs), {"k": kcol}, length=n, stream=stream)

        def gbs():
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
            agg.set_predicate(">", x_thr)
            agg.next([kg], [vcol, vcol], pred=vcol, nrows=n, stream=stream)
            state["ng"] = agg.finish(stream=stream)
        ms, sp = _measure(torch, lib, ctypes, gbs, AGG_SPANS, steps, warmup + 1)
        sweep[f"{gs:.0e}"] = {"ms_per_step": round(ms, 3), "kernels_ms": {k_: round(v_, 3) for k_, v_ in sp.items()}, "result_rows": int(state["ng"])}
        del kg
    out["group-by sweep"] = {"workload": f"SELECT k,sum(v),avg(v) WHERE v>{x_thr} GROUP BY k (hint-less) over {n:.3g} rows, keys folded to G groups",
                             "by_groups": sweep}

    # ---- configs[0]'s query shape at scale: SELECT k, count(*) GROUP BY k (no predicate, key column only)
    def gc():
        agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.COUNT_STAR, None, None)], expected_groups=7 if args.hint else 0)
        agg.next([k7], [None], nrows=n, stream=stream)
        state["ng"] = agg.finish(stream=stream)
    ms, sp = _measure(torch, lib, ctypes, gc, AGG_SPANS, steps + 2, warmup + 1)
    def gcl():
        agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.COUNT_STAR, None, None)])
        agg.next([kcol], [None], nrows=n, stream=stream)
        state["ng_l"] = agg.finish(stream=stream)
    ms_l, sp_l = _measure(torch, lib, ctypes, gcl, AGG_SPANS, steps, warmup + 1)
    out["configs[0] query shape, large G"] = _entry(f"SELECT k,count(*) GROUP BY k; N={n:.3g}, G={groups:.3g} (hint-less; the entries are bare key codes)",
                                                    n, ms_l, sp_l, 8.0 * n + 16.0 * state["ng_l"], state["ng_l"])
    out["configs[0] query shape"] = _entry(f"SELECT k,count(*) GROUP BY k; N={n:.3g}, 7 groups (the 1M-row CSV query of configs[0], at scale)",
                                           n, ms, sp, 8.0 * n + 16.0 * state["ng"], state["ng"])
    # ---- configs[4]: ORDER BY v DESC LIMIT 10 and `v*2+1, v-a, a*b` over 1e9 rows (v ~ N(11, 9) seed 2)
    gg = torch.Generator(device=device); gg.manual_seed(2)
    v4 = torch.randn(n, device=device, dtype=torch.float64, generator=gg) * 3.0 + 11.0
    v4col = DeviceColumn.from_torch(v4)

    def topk():
        state["idx"] = ops.sort_indices([v4col], [L.DESC], limit=10, stream=stream)
    ms, sp = _measure(torch, lib, ctypes, topk, [b"topk_sample", b"topk_select", b"topk_small_sort", b"radix_pass"], steps + 2, warmup + 1)
    out["configs[4] top-K"] = _entry(f"ORDER BY v DESC LIMIT 10 over {n:.3g} fp64 rows (row ids out)", n, ms, sp, 8.0 * n + 80.0, 10)

    def fullsort():
        state["idx"] = ops.sort_indices([v4col], [L.DESC], limit=0, stream=stream)
    ms, sp = _measure(torch, lib, ctypes, fullsort, [b"sort_sample", b"sort_scatter1", b"sort_scatter2", b"sort_local", b"radix_pass"], max(3, steps // 2), warmup)
    out["configs[4] full sort"] = _entry(f"ORDER BY v DESC over {n:.3g} fp64 rows, int64 row ids out (`--workload topk --limit 0`: sample sort over 8-byte entry words)",
                                         n, ms, sp, 16.0 * n, n)
    state.pop("idx", None)
    ca = torch.randn(n, device=device, dtype=torch.float64, generator=gg)
    cb = torch.rand(n, device=device, dtype=torch.float64, generator=gg)
    cols = {"v": v4col, "a": DeviceColumn.from_torch(ca), "b": DeviceColumn.from_torch(cb)}

    def proj():
        state["outs"] = ops.project_many([("add", ("mul", "v", 2), 1), ("sub", "v", "a"), ("mul", "a", "b")], cols, length=n, stream=stream)
    ms, sp = _measure(torch, lib, ctypes, proj, [b"project_kernel"], steps, warmup + 1)
    out["configs[4] projection"] = _entry(f"projection v*2+1, v-a, a*b over {n:.3g} fp64 rows (one fused kernel)", n, ms, sp, 48.0 * n, n)
    # ---- the reporting-query shape: few groups, several aggregates over several float64 columns (agg_hotn_kernel)
    acol4, bcol4 = cols["a"], cols["b"]

    def rep():
        agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()],
                                  [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64()), (L.SUM, 2, pa.float64()), (L.AVG, 3, pa.float64()),
                                   (L.COUNT_STAR, None, None)])
        agg.next([k7], [v4col, v4col, acol4, bcol4, None], nrows=n, stream=stream)
        state["rcols"] = agg.result_device(stream=stream)
        state["ng"] = agg.result_rows
    ms, sp = _measure(torch, lib, ctypes, rep, AGG_SPANS, steps, warmup + 1)
    out["reporting query, 3 columns"] = _entry(f"SELECT k,sum(v),avg(v),sum(a),avg(b),count(*) GROUP BY k; N={n:.3g}, 7 groups, three fp64 input columns "
                                               "(result columns included)", n, ms, sp, 32.0 * n + 48.0 * state["ng"], state["ng"])
    del k7, acol4, bcol4
    del v4, ca, cb, cols, v4col
    state.clear()
    # ---- BASELINE configs[3], the ONE-GPU leg: the same query over 60 HBM-resident batches of 2^24 rows streamed into ONE operator
    # (what TableReaderOperator / stream_csv hand the aggregate), G = 1e6 and G = 7
    B = 1 << 24
    nb = min(60, max(1, n // B))
    for gs, tag in ((1_000_000, "G=1e6"), (7, "G=7")):
        if gs > groups:
            continue
        kg = ops.project(("mod", "k", gs), {"k": kcol}, length=n, stream=stream)
        kt = torch.as_tensor(CudaArrayView(kg.values_ptr, n, "<i8"), device=device)
        vt = torch.as_tensor(CudaArrayView(vcol.values_ptr, n, "<f8"), device=device)
        parts = [(DeviceColumn.from_torch(kt[i * B:(i + 1) * B]), DeviceColumn.from_torch(vt[i * B:(i + 1) * B])) for i in range(nb)]

        def streamed():
            # stream_mode (vnm_agg_set_async): next() records the batch; the waiting batches go to the device as the segments of
            # one launch -- no launch / allocation / host read-back per batch
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())], stream_mode=True)
            agg.set_predicate(">", x_thr)
            for kc_, vc_ in parts:
                agg.next([kc_], [vc_, vc_], pred=vc_, nrows=B, stream=stream)
            state["cols"] = agg.result_device(stream=stream)
            state["ng"] = agg.result_rows
        ms, sp = _measure(torch, lib, ctypes, streamed, AGG_SPANS, 2, 1)
        out[f"configs[3] one-GPU leg, {tag}"] = _entry(
            f"{nb} batches x 2^24 rows (HBM-resident) streamed into one operator, result columns included; {tag}", nb * B, ms, sp,
            16.0 * nb * B + 24.0 * state["ng"], state["ng"], {"batches": nb, "ms_per_batch": ms / nb})
        if gs == 1_000_000:
            # ... and with THREE input columns: the operator records the batches, cuts the program into one part per column from the
            # stream's total rows, the parts record in turn and launch once each
            g3 = torch.Generator(device=device); g3.manual_seed(5)
            c2 = torch.randint(0, 1 << 14, (nb * B,), device=device, dtype=torch.int64, generator=g3).to(torch.float64) / 128.0
            c3 = torch.randint(0, 1 << 14, (nb * B,), device=device, dtype=torch.int64, generator=g3).to(torch.float64) / 128.0
            parts3 = [(p_[0], p_[1], DeviceColumn.from_torch(c2[i * B:(i + 1) * B]), DeviceColumn.from_torch(c3[i * B:(i + 1) * B])) for i, p_ in enumerate(parts)]

            def streamed3():
                agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()],
                                          [(L.SUM, 1, pa.float64()), (L.AVG, 2, pa.float64()), (L.SUM, 3, pa.float64()), (L.COUNT_STAR, None, None)], stream_mode=True)
                agg.set_predicate(">", x_thr)
                for kc_, a_, b_, c_ in parts3:
                    agg.next([kc_], [a_, b_, c_, None], pred=a_, nrows=B, stream=stream)
                state["cols"] = agg.result_device(stream=stream)
                state["ng"] = agg.result_rows
            ms, sp = _measure(torch, lib, ctypes, streamed3, AGG_SPANS + [b"agg_split_join"], 2, 1)
            out["configs[3] one-GPU leg, G=1e6, three input columns"] = _entry(
                f"{nb} batches x 2^24 rows streamed into one operator: SELECT k,sum(a),avg(b),sum(c),count(*) WHERE a>{x_thr} GROUP BY k; G=1e6, result columns included",
                nb * B, ms, sp, 32.0 * nb * B + 48.0 * state["ng"], state["ng"], {"batches": nb})
            del parts3, c2, c3
        del parts, kt, vt, kg
    state.clear()
    # ---- SURVEY.md 8d's second line: END TO END from HOST memory -- Arrow record batches in pageable host memory through the
    # Arrow-level boundary (vnm_agg_op_next / vnm_agg_op_result = vinum_lib.SingleNumericalHashAggregate.next / .result):
    # pinned double-buffered staging -> HBM -> fused filter-less aggregate -> Arrow result on the host.  2^24-row batches.
    try:
        from vinum_amd import vinum_lib as vl
        ne = min(n, 1 << 27)
        rng = np.random.default_rng(7)
        hk = rng.integers(0, 1000, ne).astype(np.int64)
        hv = rng.integers(0, 1 << 14, ne).astype(np.float64) / 128.0
        hb = pa.table({"k": hk, "v": hv}).to_batches(max_chunksize=B)
        best = None
        for rep in range(3):
            agg = vl.SingleNumericalHashAggregate(["k"], ["k"], [vl.AggFuncDef(vl.SUM, "v", "s"), vl.AggFuncDef(vl.AVG, "v", "a")])
            t0 = time.perf_counter()
            for b in hb:
                agg.next(b)
            res = agg.result()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["h2d_end_to_end"] = {"workload": f"SELECT k,sum(v),avg(v) GROUP BY k over {ne:.3g} HOST-resident rows (pageable Arrow buffers, 1000 groups) in "
                                             f"2^24-row record batches through vnm_agg_op_next / vnm_agg_op_result, Arrow result on the host",
                                 "rows_per_s": ne / best, "ms": best * 1e3, "pcie_GB_per_s": 16.0 * ne / best / 1e9, "result_rows": res.num_rows,
                                 "note": "PCIe-inclusive: never the bench value (inputs of `value` are resident in HBM)"}
        # ... and in the reference's default batch size (10 000 rows, vinum/__init__.py:52): the wrapper keeps small batches and
        # hands them over as one Arrow C stream; the library stages the chunks through the pinned ring without joining them
        sm = pa.table({"k": hk[:50_000_000], "v": hv[:50_000_000]}).to_batches(max_chunksize=10_000)
        best_s = None
        for rep in range(3):
            agg = vl.SingleNumericalHashAggregate(["k"], ["k"], [vl.AggFuncDef(vl.SUM, "v", "s"), vl.AggFuncDef(vl.AVG, "v", "a")])
            t0 = time.perf_counter()
            for b in sm:
                agg.next(b)
            res_s = agg.result()
            dt = time.perf_counter() - t0
            best_s = dt if best_s is None else min(best_s, dt)
        ns = sum(b.num_rows for b in sm)
        out["h2d_small_batches"] = {"workload": f"the same query over {ns:.3g} host rows in {len(sm)} record batches of 10 000 rows (the reference's default batch size)",
                                    "rows_per_s": ns / best_s, "ms": best_s * 1e3, "result_rows": res_s.num_rows,
                                    "note": "PCIe-inclusive; r02: 0.55 Grows/s (host-side concatenation of the waiting batches)"}
        del hb, hk, hv, sm
    except Exception as e:
        out["h2d_end_to_end"] = {"error": str(e)}
    # ---- f3: GROUP BY a STRING key from host memory (GenericHashAggregate: the key column is dictionary-encoded on the device,
    # vnm_strdict_encode, then the int32 codes go through the numeric operator); the host route this replaced (Arrow's
    # dictionary_encode + a NumPy merge) is timed beside it on a tenth of the rows
    try:
        from vinum_amd import vinum_lib as vl
        ns = min(n, 1 << 25)
        rng = np.random.default_rng(9)
        cities = pa.array([f"city_{i:06d}" for i in range(100_000)])
        sk = cities.take(pa.array(rng.integers(0, 100_000, ns)))
        sv = pa.array(rng.integers(0, 1 << 14, ns).astype(np.float64) / 128.0)
        sb = pa.table({"city": sk, "v": sv}).to_batches(max_chunksize=1 << 22)
        best = None
        for rep in range(3):
            agg = vl.GenericHashAggregate(["city"], ["city"], [vl.AggFuncDef(vl.SUM, "v", "s"), vl.AggFuncDef(vl.COUNT_STAR, "", "n")])
            t0 = time.perf_counter()
            for b in sb:
                agg.next(b)
            res = agg.result()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        kd = vl.KeyDictionary(pa.string())
        kd._device = False
        sub = sk.slice(0, ns // 10)
        kd.encode(sub)
        t0 = time.perf_counter(); kd.encode(sub); host_dt = time.perf_counter() - t0
        out["string_key_groupby"] = {"workload": f"SELECT city,sum(v),count(*) GROUP BY city over {ns:.3g} HOST-resident rows, 1e5 distinct 11-byte strings, "
                                                 f"2^22-row record batches through vinum_lib.GenericHashAggregate (device string dictionary)",
                                     "rows_per_s": ns / best, "ms": best * 1e3, "result_rows": res.num_rows,
                                     "host_dictionary_encode_rows_per_s": len(sub) / host_dt,
                                     "note": "PCIe-inclusive; the host figure is the key column's dictionary encoding ALONE (the r02 route)"}
        del sb, sk, sv
    except Exception as e:
        out["string_key_groupby"] = {"error": str(e)}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)          # (does not return)
    if os.environ.get("VNM_BENCH_DRY_RUN") == "1":
        return dry_run(args)
    # RCCL / HIP runtime banners are written to fd 1 by native code: keep the real stdout for the ONE JSON line
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # VNM_BENCH_SHARED_GPU=1: every rank on cuda:0 with the gloo backend (RCCL refuses two ranks on one device) -- a functional run
    # of the N-rank code path on a one-GPU box (tests/test_gpu_bench_check.py); never a measurement
    shared_gpu = os.environ.get("VNM_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # VNM_BENCH_FORCE_EXCHANGE=1 runs the RCCL exchange + merge step with a single rank too (self-test of the
    # N > 1 code path on a 1-GPU box)
    force_exchange = os.environ.get("VNM_BENCH_FORCE_EXCHANGE") == "1"
    if world > 1 or force_exchange:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)

    from vinum_amd import _lib as L
    from vinum_amd import ops
    from vinum_amd.device import DeviceColumn
    import pyarrow as pa
    lib = L.lib()

    n = int(args.rows)
    groups = int(args.groups)
    x_thr = threshold_for(args.selectivity)
    B = 1 << 24
    if args.workload == "stream":      # configs[3]: this rank's share of the stream, 2^24-row record batches resident in HBM
        nb = args.batches if args.batches > 0 else min(60, max(1, n // B))
        n = nb * B
    k, v = gen_data(torch, n, groups, seed=1 + rank, device=device)
    kcol = DeviceColumn.from_torch(k)
    vcol = DeviceColumn.from_torch(v)
    stream = torch.cuda.current_stream().cuda_stream

    def batches_of(kt, vt):
        return [(DeviceColumn.from_torch(kt[i * B:(i + 1) * B]), DeviceColumn.from_torch(vt[i * B:(i + 1) * B])) for i in range(kt.numel() // B)]
    # what step() runs: the headline case first, the `also` cases of a multi-rank run afterwards
    cur = {"kind": args.workload, "kcol": kcol, "vcol": vcol, "parts": batches_of(k, v) if args.workload == "stream" else None, "hint": groups}
    out_buf = None
    if args.workload == "filter":
        out_buf = torch.empty(n, dtype=torch.float64, device=device)
    if args.workload in ("topk", "project"):
        # configs[4]: v ~ N(11, 9) seed 2, two more fp64 columns for `v*2+1, v-a, a*b`
        gg = torch.Generator(device=device); gg.manual_seed(2 + rank)
        del k
        v = torch.randn(n, device=device, dtype=torch.float64, generator=gg) * 3.0 + 11.0
        vcol = DeviceColumn.from_torch(v)
        if args.workload == "project":
            ca = torch.randn(n, device=device, dtype=torch.float64, generator=gg)
            cb = torch.rand(n, device=device, dtype=torch.float64, generator=gg)
            acol, bcol = DeviceColumn.from_torch(ca), DeviceColumn.from_torch(cb)

    state = {}

    def step():
        if args.workload == "filter":
            ov = (ctypes.c_void_p * 1)(out_buf.data_ptr())
            ob = (ctypes.c_void_p * 1)(None)
            cnt = ctypes.c_int64(0)
            d = vcol.dcol()
            L.check(lib.vnm_filter_cmp(ctypes.byref(d), L.GT, 1, x_thr, 0, 1, ctypes.byref(d), ov, ob,
                                       ctypes.byref(cnt), ctypes.c_void_p(stream)))
            state["out_rows"] = cnt.value
            return
        if args.workload == "topk":
            idx = ops.sort_indices([vcol], [L.DESC], limit=args.limit, stream=stream)
            state["out_rows"] = args.limit
            state["idx"] = idx
            return
        if args.workload == "project":
            cols = {"v": vcol, "a": acol, "b": bcol}
            state["outs"] = ops.project_many([("add", ("mul", "v", 2), 1), ("sub", "v", "a"), ("mul", "a", "b")],
                                             cols, length=n, stream=stream)
            state["out_rows"] = n
            return
        if cur["kind"] == "stream":
            # BaseAggregate::Next per record batch (base_aggregate.cpp:23-45) into ONE operator; stream_mode: next() records the
            # batch, the waiting batches go to the device as the segments of one launch (vnm_agg_set_async)
            dist_mode = world > 1 or force_exchange
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())],
                                      expected_groups=cur["hint"] if args.hint else 0, rank_aligned=dist_mode, stream_mode=True)
            agg.set_predicate(">", x_thr)
            parts = cur["parts"]
            if dist_mode:
                from vinum_amd import distributed as D
                # the ranks agree on ONE group-count estimate and ONE key range (a single all_gather): every operator then cuts its
                # result the same way and the dense path's tables are slot-compatible.  ONCE per standing query (ExchangePlan): the
                # steps after the first apply what was agreed without a collective
                if os.environ.get("VNM_BENCH_EXCHANGE", "dense") == "dense":
                    plans = state.setdefault("plans", {})
                    pk = ("stream", cur["hint"], len(parts))
                    if pk not in plans:
                        plans[pk] = D.agree_on_plan(agg, parts[0][0], B, device, stream=stream, estimate=not args.hint, hint=cur["hint"])
                    else:
                        plans[pk].apply(agg, use_estimate=not args.hint)
                    state["plan"] = plans[pk]
                elif not args.hint:
                    D.agree_on_group_count(agg, parts[0][0], B, device, stream=stream)
            for kc_, vc_ in parts:
                agg.next([kc_], [vc_, vc_], pred=vc_, nrows=B, stream=stream)
            if dist_mode:
                ng = exchange_and_merge(agg, None)
            else:
                state["cols"] = agg.result_device(stream=stream)
                ng = agg.result_rows
            state["out_rows"] = ng
            state.pop("agg", None)
            state["agg"] = agg
            return
        gk, gv = cur["kcol"], cur["vcol"]   # (names of their own: `vcol` is the closure variable of the other workloads)
        if args.shape == "count_star":
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.COUNT_STAR, 0, None)],
                                      expected_groups=groups if args.hint else 0)
            agg.next([gk], [None], nrows=n, stream=stream)
        elif args.shape == "minmax":
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()],
                                      [(L.MIN, 1, pa.float64()), (L.MAX, 1, pa.float64())],
                                      expected_groups=groups if args.hint else 0)
            agg.set_predicate(">", x_thr)
            agg.next([gk], [gv, gv], pred=gv, nrows=n, stream=stream)
        else:
            agg = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()],
                                      [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())],
                                      expected_groups=cur["hint"] if args.hint else 0, rank_aligned=(world > 1 or force_exchange))
            agg.set_predicate(">", x_thr)
            if world > 1 or force_exchange:
                # several ranks: agree on ONE group-count estimate (or ranks may cut their results into different numbers of
                # partitions) and on ONE key range (the dense-key path then gives slot-compatible tables on every rank) -- one
                # all_gather, once per standing query (distributed.ExchangePlan)
                from vinum_amd import distributed as D
                if os.environ.get("VNM_BENCH_EXCHANGE", "dense") == "dense":
                    plans = state.setdefault("plans", {})
                    pk = ("groupby", cur["hint"], n)
                    if pk not in plans:
                        plans[pk] = D.agree_on_plan(agg, gk, n, device, stream=stream, estimate=not args.hint, hint=cur["hint"])
                    else:
                        plans[pk].apply(agg, use_estimate=not args.hint)
                    state["plan"] = plans[pk]
                elif not args.hint:
                    D.agree_on_group_count(agg, gk, n, device, stream=stream)
            agg.next([gk], [gv, gv], pred=gv, nrows=n, stream=stream)
        if world > 1 or force_exchange:
            ng = exchange_and_merge(agg, None)
        else:
            # BaseAggregate::Result: the result COLUMNS (key, sum, avg as Arrow-layout buffers in HBM) are part of the step
            state["cols"] = agg.result_device(stream=stream)
            ng = agg.result_rows
        state["out_rows"] = ng
        state.pop("agg", None)
        state["agg"] = agg  # keep the last result alive for the sanity check; previous one is freed

    def exchange_and_merge(agg, ng):
        """Key-partitioned exchange of partial groups (owner = mix(key words) mod P): the run is bucketed by owner
        on the device, ONE all_to_all over RCCL moves it, the owner merges what it received
        (vinum_amd/distributed.py)."""
        from vinum_amd import distributed as D
        kw, aw = agg.layout()
        want = os.environ.get("VNM_BENCH_EXCHANGE", "dense")
        make0 = lambda: ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()], [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
        plan = state.get("plan")
        if want == "dense" and plan is not None and plan.route == "small":
            # few groups (the agreed estimate says so): ONE fixed-size all_gather, merged on the device from the blocks' own headers --
            # no route agreement, no count exchange, no slicing (distributed.exchange_small_fixed)
            ng0 = agg.finish(stream=stream)      # (the aggregation itself: the waiting batches of the stream go to the device here)
            torch.cuda.synchronize()
            t_a = time.perf_counter()
            send = torch.empty((max(ng0, 1), kw + aw), dtype=torch.int64, device=device)
            agg.bucket_by_owner(1, send.data_ptr(), stream=stream)

            def merge_blocks(blocks, counts):
                m = make0()
                m.merge_row_blocks(int(blocks.shape[0]), int(blocks.shape[1]) - 1, blocks.data_ptr(), stream=stream)
                m._keep = blocks
                return m
            small, _counts = D.exchange_small_fixed(send[:ng0], merge_blocks)
            if small is not None:
                out = small.finish(stream=stream)
                torch.cuda.synchronize()
                ph = state.setdefault("phases", {"bucket": 0.0, "all_to_all": 0.0, "merge": 0.0, "partition_aligned": 0.0})
                ph["small_fixed"] = ph.get("small_fixed", 0.0) + (time.perf_counter() - t_a) * 1e3
                state["merged"] = small
                state["replicated"] = True
                state["exchange_kind"] = "small_fixed"
                return out
        # the last pass of the aggregation itself (the dense path's deferred final pass, here writing its direct-addressed tables)
        got = agg.dense_table(stream=stream) if want == "dense" else None
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        make = lambda: ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()],
                                           [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())])
        # large G through the dense-key path with an agreed code range: the final pass's direct-addressed tables travel as
        # they are (equal known splits, no bucketing) and add up slot by slot on their owner
        merged = D.exchange_dense(agg, make, device, stream=stream, got=got) if want == "dense" else None
        if merged is not None:
            out = merged.finish(stream=stream)
            torch.cuda.synchronize()
            ph = state.setdefault("phases", {"bucket": 0.0, "all_to_all": 0.0, "merge": 0.0, "partition_aligned": 0.0, "dense_tables": 0.0})
            ph["dense_tables"] = ph.get("dense_tables", 0.0) + (time.perf_counter() - t_a) * 1e3
            state["merged"] = merged
            state["exchange_kind"] = "dense_tables"
            return out
        ng = agg.finish(stream=stream)
        # large G otherwise: partition-aligned exchange (owners merge hash partitions in LDS, no HBM atomics)
        merged = D.exchange_partition_aligned(agg, make, device) if want in ("dense", "aligned") else None
        if merged is not None:
            out = merged.finish(stream=stream)
            torch.cuda.synchronize()
            ph = state.setdefault("phases", {"bucket": 0.0, "all_to_all": 0.0, "merge": 0.0, "partition_aligned": 0.0})
            ph["partition_aligned"] += (time.perf_counter() - t_a) * 1e3
            state["merged"] = merged
            state["exchange_kind"] = "partition_aligned"
            return out
        send = torch.empty((max(ng, 1), kw + aw), dtype=torch.int64, device=device)
        counts = agg.bucket_by_owner(world, send.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        # small result sets: one all_gather, every rank merges all partial rows (no all_to_all); the merged result is
        # replicated, so only rank 0's copy counts as output rows
        def merge_all(allrows):
            m = make()
            m.merge_rows(int(allrows.shape[0]), allrows.data_ptr(), stream=stream)
            m._keep = allrows
            return m
        small = D.exchange_allgather_small(send[:ng], merge_all)
        if small is not None:
            out = small.finish(stream=stream)
            torch.cuda.synchronize()
            ph = state.setdefault("phases", {"bucket": 0.0, "all_to_all": 0.0, "merge": 0.0, "partition_aligned": 0.0, "allgather_small": 0.0})
            ph["allgather_small"] = ph.get("allgather_small", 0.0) + (time.perf_counter() - t_a) * 1e3
            state["merged"] = small
            state["replicated"] = True
            state["exchange_kind"] = "allgather_small"
            return out

        def merge(recv):
            torch.cuda.synchronize()
            t_c = time.perf_counter()
            merged = ops.DeviceAggregate(L.SINGLE_NUMERICAL, [pa.int64()],
                                         [(L.SUM, 1, pa.float64()), (L.AVG, 1, pa.float64())],
                                         expected_groups=max(int(recv.shape[0]), 1024))
            merged.merge_rows(int(recv.shape[0]), recv.data_ptr(), stream=stream)
            out = merged.finish(stream=stream)
            torch.cuda.synchronize()
            ph = state.setdefault("phases", {"bucket": 0.0, "all_to_all": 0.0, "merge": 0.0, "partition_aligned": 0.0})
            ph["bucket"] += (t_b - t_a) * 1e3
            ph["all_to_all"] += (t_c - t_b) * 1e3
            ph["merge"] += (time.perf_counter() - t_c) * 1e3
            state["merged"] = (merged, recv)
            state["exchange_kind"] = "bucketed"
            return out

        return D.exchange_bucketed(send[:ng], counts, merge)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(steps, warmup):
        """`warmup` untimed steps, then exactly `steps` steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
        for _ in range(warmup):
            step()
        sync_all()
        state.pop("phases", None)
        lib.vnm_set_profiling(1)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync_all()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed = timed(args.steps, args.warmup)
    headline_phases = dict(state["phases"]) if "phases" in state else None     # (the `also` cases of a multi-rank run reuse `state`)
    headline_exchange = state.get("exchange_kind")

    check = None
    # The property check of the LAST step's result (after the timed region, on the device, reduced over all ranks) is part of every
    # hot-shape line: survivors and totals conserved, (one GPU) the result columns equal to the operator's partial state.
    # --check: a failed property aborts the run; otherwise it is reported in the line ("ok": false).  --no-check skips it.
    if not args.no_check and args.workload in ("groupby", "stream") and args.shape == "hot":
        try:
            check = check_groupby(torch, dist, state, k, v, x_thr, world, rank, device, force_exchange, strict=bool(args.check))
        except AssertionError:
            raise
        except Exception as e:      # noqa: BLE001 -- (out of memory for the torch reference ...): the measurement stands, the check is reported as not run
            if args.check:
                raise
            check = {"ok": None, "error": repr(e)}
        torch.cuda.empty_cache()    # (the check's torch temporaries: the side measurements need the room)
    names = {"filter": [b"filter_kernel"], "topk": [b"topk_sample", b"topk_select", b"topk_small_sort", b"sort_encode", b"radix_hist", b"radix_pass",
                                                    b"sort_sample", b"sort_scatter1", b"sort_scatter2", b"sort_local"], "project": [b"project_kernel"]}.get(
        args.workload, AGG_SPANS)
    spans = {}
    for nm in names:
        tot_ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        lib.vnm_profile_query(nm, ctypes.byref(tot_ms), ctypes.byref(cnt))
        if cnt.value:
            spans[nm.decode()] = (tot_ms.value / max(args.steps, 1), cnt.value / max(args.steps, 1))
    lib.vnm_set_profiling(0)
    # the dominant kernel = the one with the largest time per step (HIP events on the launch stream)
    dom_name = max(spans, key=lambda k: spans[k][0]) if spans else "none"
    kernel_ms = spans[dom_name][0] if spans else 0.0
    cnt = ctypes.c_int64(int(round(spans[dom_name][1] * args.steps))) if spans else ctypes.c_int64(0)

    out_rows = state["out_rows"]
    if args.workload == "filter":
        alg_bytes = 8.0 * n + 8.0 * out_rows          # read column once, write survivors
        workload = f"configs[1]: WHERE fare_amount > {x_thr} over {n:.3g}-row fp64 column (s={args.selectivity}) -> compacted column"
        dom = "filter_kernel"
    elif args.workload == "topk":
        alg_bytes = 8.0 * n + 8.0 * args.limit       # SURVEY.md §8d config 5, top-K variant
        workload = f"configs[4]: ORDER BY v DESC LIMIT {args.limit} over {n:.3g} fp64 rows (top-K variant, row ids out)"
        dom = "topk_select"
        if args.limit == 0:                          # full-sort variant: read keys, write the sorted order (int64 row ids)
            alg_bytes = 16.0 * n
            workload = f"configs[4]: ORDER BY v DESC over {n:.3g} fp64 rows (full stable sort -- sample sort over 8-byte entry words --, int64 row ids out)"
    elif args.workload == "project":
        alg_bytes = 8.0 * n * 3 + 8.0 * n * 3       # three distinct inputs, three outputs
        workload = f"configs[4]: projection v*2+1, v-a, a*b over {n:.3g} fp64 rows"
        dom = "project_kernel"
    else:
        alg_bytes = 16.0 * n + 24.0 * out_rows        # read key+value once, write key,sum,avg per group
        workload = (f"configs[2]: SELECT k,sum(v),avg(v) WHERE v>{x_thr} GROUP BY k; N={n:.3g} rows/GPU, "
                    f"G={groups:.3g} int64 keys, s={args.selectivity}")
        if args.shape == "count_star":
            alg_bytes = 8.0 * n + 16.0 * out_rows
            workload = f"SELECT k,count(*) GROUP BY k (configs[0]'s query shape); N={n:.3g} rows/GPU, G={groups:.3g}"
        elif args.shape == "minmax":
            workload = (f"SELECT k,min(v),max(v) WHERE v>{x_thr} GROUP BY k; N={n:.3g} rows/GPU, G={groups:.3g}, "
                        f"s={args.selectivity}")
        if args.workload == "stream":
            workload = (f"configs[3]: SELECT k,sum(v),avg(v) WHERE v>{x_thr} GROUP BY k over a stream of {n // B} x 2^24-row record batches per GPU "
                        f"(HBM-resident, dealt round-robin to {world} rank{'s' if world > 1 else ''}), G={groups:.3g}, s={args.selectivity}; "
                        + ("partial aggregates exchanged over RCCL inside the step" if world > 1 or force_exchange else "result columns inside the step"))
    if args.workload not in ("filter",):
        dom = " + ".join(spans) + f" (dominant: {dom_name})" if len(spans) > 1 else dom_name
    # per-kernel algorithmic bytes: the scan kernels read key+value once (16 N) and write the groups; a
    # partition pass of the large-G path reads 16 N and its successors re-read the surviving pairs -- the
    # roofline of THE QUERY is always computed from the query's algorithmic bytes over the SUM of its kernels
    total_kernel_ms = sum(v[0] for v in spans.values()) if spans else 0.0
    achieved = alg_bytes / (total_kernel_ms * 1e-3) / 1e9 if total_kernel_ms > 0 else 0.0

    traffic = None
    traffic_src = None
    # HBM bytes per step from the PMC passes of the SAME command (tools/profile.sh -> profiles/rNN_rocprofv3_pmc_*.txt),
    # newest round first; hinted and hint-less runs take different paths, so they have different entries
    key = (f"{args.workload}_N{n:.0e}_G{groups:.0e}_s{args.selectivity}" + (f"_{args.shape}" if args.shape != "hot" else "")
           + ("_hint" if args.hint else ""))      # (the keys tools/traffic_from_pmc.py writes: other shapes run other kernels)
    for tf in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", tf)) as f:
                tj = json.load(f)
        except Exception:
            continue
        k2 = key if key in tj else (key[:-5] if tf == "r01_traffic.json" and args.hint and key[:-5] in tj else None)
        if k2:
            ent = tj[k2]
            # pass 1 runs in one of two states, decided per process (profiles/tuning_log_r02.md): take the PMC profile whose pass-1 duration matches
            # the one measured live
            live = spans.get("agg_part_scatter1", (None,))[0]
            cands = [ent] + list(ent.get("variants", []))
            if live is not None and all("kernel_ms" in c for c in cands):
                ent2 = min(cands, key=lambda c: abs(c["kernel_ms"].get("agg_part_scatter1", 0.0) - live))
            else:
                ent2 = ent
            traffic = ent2["bytes_per_step"]
            traffic_src = ent2.get("source", "profiles/") + " (" + ent.get("how", "") + ")"
            break
    # ---- a multi-rank run's other configurations (every rank takes part: the exchanges are collectives), a few steps each, same
    # bracketing: configs[3] at G = 7 (all-gather of a handful of partial groups) and the one-batch G = 1e8 shape of configs[2]
    # (dense tables over all_to_all)
    multi_also = None
    if (world > 1 or force_exchange) and args.workload == "stream" and not args.no_also:
        multi_also = {}
        try:
            def also_entry(tag, what, nrows):
                exch0 = dict(state.get("phases", {}))
                el = timed(2, 1)
                sp = _spans(lib, ctypes, AGG_SPANS, 2)
                lib.vnm_set_profiling(0)
                multi_also[tag] = {"workload": what, "rows_per_s": nrows * world * 2 / el, "ms_per_step": el / 2 * 1e3, "result_rows": int(state["out_rows"]),
                                   "exchange": state.get("exchange_kind"),
                                   "exchange_ms_per_step": {k2: round(v2 / 2, 3) for k2, v2 in state.get("phases", {}).items()},
                                   "kernels_ms": {k2: round(v2, 4) for k2, v2 in sp.items()}}
            cur.update(kind="stream", parts=batches_of(torch.remainder(k, 7), v), hint=7)
            also_entry("configs[3] stream, G=7", f"the same stream with 7 groups ({n // B} x 2^24-row batches per rank)", n)
            cur["parts"] = None
            state.clear()
            gg8 = torch.Generator(device=device); gg8.manual_seed(101 + rank)
            k8 = torch.randint(0, 10**8, (n,), device=device, dtype=torch.int64, generator=gg8)
            cur.update(kind="groupby", kcol=DeviceColumn.from_torch(k8), vcol=vcol, hint=10**8)
            also_entry("configs[2] shape, G=1e8", f"ONE batch of {n:.3g} rows per rank, 1e8 groups: dense tables exchanged by all_to_all", n)
            state.clear()
            del k8
        except Exception as e:
            multi_also["error"] = str(e)
    if rank == 0:
        result = {
            "metric": "rows/sec + achieved HBM GB/s, filter->group-by over 10^9-row Arrow batches",
            "value": n * world * args.steps / elapsed,
            "unit": "rows/s",
            "n_gpus": world, "rccl_ranks": dist.get_world_size() if (world > 1 or force_exchange) else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64/int64", "data": "synthetic" + (" (FUNCTIONAL RUN: all ranks share one GPU over gloo; not a measurement)" if shared_gpu else ""),
            "config": {"workload": workload, "rows_per_gpu": n, "groups": groups if args.workload == "groupby" else None,
                       "selectivity": args.selectivity, "result_rows": int(out_rows),
                       "group_count_hint": "given (vnm_agg_set_hint)" if args.hint else "none: the operator samples the keys (what the reference boundary allows)",
                       "parallelism": f"batch-sharded x{world}" + (", RCCL all_to_all partial-aggregate exchange" if world > 1 else ""),
                       "scaling_reference": ("the SAME workload on one GPU is also['configs[3] one-GPU leg, G=1e6'] of the --gpus 1 line "
                                             "(its default workload is configs[2], one 1e9-row batch at G = 1e8)") if args.workload == "stream" and world > 1 else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": dom,
                         "kernel_ms": total_kernel_ms, "dominant_kernel_ms": kernel_ms,
                         "kernels_ms": {k: round(v[0], 4) for k, v in spans.items()},
                         "algorithmic_bytes": alg_bytes,
                         "launches_per_step": cnt.value / max(args.steps, 1)},
            "exchange_ms_per_step": ({k2: round(v2 / max(args.steps, 1), 3) for k2, v2 in headline_phases.items()}
                                     if headline_phases is not None else None),
            "exchange": headline_exchange,
        }
        if check is not None:
            result["check"] = check
        #  the device's state next to the line, and a SUSTAINED leg: the same step for a few seconds after the timed
        # region -- the timed K steps are ~0.2 s of GPU work in a minute of data generation, invisible to a 5-s utilisation sampler and
        # too short to show what the rate settles at; reported beside the headline, never part of `value`
        result["device_state"] = device_state()
        if world == 1 and not force_exchange and not args.no_also and args.sustain_seconds > 0:
            try:
                n_s, t_s = 0, time.perf_counter()
                while time.perf_counter() - t_s < args.sustain_seconds:
                    for _ in range(10):
                        step()
                    torch.cuda.synchronize()
                    n_s += 10
                el_s = time.perf_counter() - t_s
                result["sustained"] = {"seconds": round(el_s, 2), "steps": n_s, "ms_per_step": el_s / n_s * 1e3, "device_state_after": device_state()}
            except Exception as e:  # noqa: BLE001 -- reporting only
                result["sustained"] = {"error": str(e)}
        if multi_also is not None:
            result["also"] = multi_also
        if world == 1 and not force_exchange and args.workload == "groupby" and args.shape == "hot" and not args.no_also:
            # the other single-GPU configurations of BASELINE.json on the same resident column, a few steps each
            # (reported beside the headline; not part of `value`)
            try:
                result["also"] = side_workloads(torch, lib, L, ops, pa, ctypes, kcol, vcol, n, x_thr, stream, args)
            except Exception as e:
                result["also"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline and args.workload in ("groupby", "filter") and args.shape == "hot":
            try:
                result["cpu_baseline"] = cpu_baseline(args, x_thr)
            except Exception as e:  # the baseline is reporting only; never fail the bench line on it
                result["cpu_baseline"] = {"value": None, "unit": "rows/s", "cores": 1, "kind": "port",
                                          "sample": f"failed: {e}"}
    else:
        result = None
    if world > 1 or force_exchange:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if result is not None:
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    os.close(real_stdout)
    if any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ):
        return   # under rocprofv3: its tool library writes the trace from an atexit handler
    os._exit(0)   # skip native atexit chatter; everything is flushed and the process group is destroyed


if __name__ == "__main__":
    main()
