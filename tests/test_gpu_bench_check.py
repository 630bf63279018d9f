"""bench.py --check through every multi-GPU exchange route with ONE rank (VNM_BENCH_FORCE_EXCHANGE=1: the RCCL collectives run
against the rank itself), at sizes that reach the large-batch paths: the property checks of the last step (survivors and totals
conserved, every key on one owner) must hold for the dense-table, partition-aligned, owner-bucketed and all-gather exchanges,
and for the single-GPU headline with its fused result columns."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *args):
    env = dict(os.environ)
    env.update(extra_env)
    env.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-also", "--no-cpu-baseline",
                        "--check", *args], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("route,env,args", [
    ("dense_tables", {"VNM_BENCH_FORCE_EXCHANGE": "1"}, ["--rows", "3e7", "--groups", "1e7"]),
    ("partition_aligned", {"VNM_BENCH_FORCE_EXCHANGE": "1", "VNM_BENCH_EXCHANGE": "aligned"}, ["--rows", "3e7", "--groups", "1e7"]),
    ("bucketed", {"VNM_BENCH_FORCE_EXCHANGE": "1", "VNM_BENCH_EXCHANGE": "bucketed"}, ["--rows", "3e7", "--groups", "1e7"]),
    # (G = 1e5 takes the dense tables too since round 4 -- the split final pass writes them; "allgather": skip the table routes)
    ("allgather_small", {"VNM_BENCH_FORCE_EXCHANGE": "1", "VNM_BENCH_EXCHANGE": "allgather"}, ["--rows", "3e7", "--groups", "1e5"]),
    ("dense_tables", {"VNM_BENCH_FORCE_EXCHANGE": "1"}, ["--rows", "3e7", "--groups", "1e5"]),
    ("dense_tables", {"VNM_BENCH_FORCE_EXCHANGE": "1"}, ["--rows", "3e7", "--groups", "3e6"]),
])
def test_bench_check_exchange_routes(route, env, args):
    j = _bench(env, *args)
    assert j["check"]["exchange"] == route, j["check"]
    assert j["check"]["survivors_conserved"] and j["check"]["totals_conserved"]
    assert j["exchange_ms_per_step"] is not None


def test_bench_check_single_gpu_headline_with_result_columns():
    j = _bench({}, "--rows", "3e7", "--groups", "1e7")
    assert j["check"]["survivors_conserved"] and j["check"]["totals_conserved"]
    assert j["check"].get("result_columns_match") is True


@pytest.mark.parametrize("groups,route", [("1e6", "dense_tables"), ("7", "small_fixed")])
def test_bench_check_stream_workload_with_exchange(groups, route):
    """configs[3] (the default of a multi-rank run): a stream of 2^24-row batches into one operator in stream mode, partial aggregates
    exchanged -- here by ONE rank with itself, `also` cases included (they are collectives on every rank)."""
    env = dict(os.environ)
    env.update({"VNM_BENCH_FORCE_EXCHANGE": "1", "MASTER_PORT": str(29900 + os.getpid() % 90)})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stream", "--batches", "4", "--groups", groups, "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--check"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    j = json.loads(p.stdout.strip().splitlines()[-1])
    assert j["check"]["exchange"] == route, j["check"]
    assert j["check"]["survivors_conserved"] and j["check"]["totals_conserved"]
    assert j["rccl_ranks"] == 1 and "configs[3]" in j["config"]["workload"]
    assert "error" not in j["also"], j["also"]
    assert j["also"]["configs[3] stream, G=7"]["exchange"] == "small_fixed"     # ONE fixed-size collective (round 5)
    assert j["also"]["configs[2] shape, G=1e8"]["exchange"] == "dense_tables"


def test_bench_check_stream_workload_single_gpu():
    j = _bench({}, "--workload", "stream", "--batches", "5", "--groups", "1e6")
    assert j["check"]["survivors_conserved"] and j["check"]["totals_conserved"]
    assert j["check"].get("result_columns_match") is True
    assert j["roofline"]["launches_per_step"] <= 1.0     # ONE launch of the scatter pass over the five waiting batches


@pytest.mark.parametrize("ranks,groups,route", [(2, "1e6", "dense_tables"), (3, "7", "small_fixed"), (2, "2e7", "dense_tables")])
def test_bench_self_launch_ranks_sharing_the_gpu(ranks, groups, route):
    """The N-rank code path for real ("a 2-process single-GPU-shared ... run of the self-launch path"): `bench.py --gpus N`
    starts N processes by itself; VNM_BENCH_SHARED_GPU=1 puts every rank on cuda:0 with the gloo backend (RCCL refuses two ranks on one
    device; device tensors are staged through the host inside the collectives).  configs[3]: every rank streams its own batches,
    the ranks agree on the group count and the code range, exchange their partial aggregates and --check reduces the properties
    over all ranks (survivors and totals conserved, no group on two ranks)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VNM_BENCH_SHARED_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--batches", "3", "--groups", groups, "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-also", "--check"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == ranks and j["rccl_ranks"] == ranks and "configs[3]" in j["config"]["workload"]
    assert j["check"]["world_size"] == ranks and j["check"]["exchange"] == route, j["check"]
    assert j["check"]["survivors_conserved"] and j["check"]["totals_conserved"]
    assert j["value"] > 0 and j["config"]["rows_per_gpu"] == 3 << 24


@pytest.mark.parametrize("args,metric", [
    (["--workload", "filter"], "filter"), (["--workload", "topk"], "topk"), (["--workload", "topk", "--limit", "0"], "sort"),
    (["--workload", "project"], "project"), (["--workload", "groupby", "--shape", "count_star", "--groups", "1e6"], "groupby"),
    (["--workload", "groupby", "--shape", "minmax", "--groups", "1e5"], "groupby")])
def test_bench_side_workloads_run(args, metric):
    """Every workload the profiles of a round are taken from (tools/profile.sh) still produces its JSON line: the r04 profiles of
    filter / top-K / full sort / projection were EMPTY because a local name in step() shadowed the other workloads' value column."""
    j = _bench({}, "--rows", "2e7", *args)
    assert j["value"] > 0 and j["ms_per_step"] > 0, j
    assert j["roofline"]["achieved"] > 0, j["roofline"]
