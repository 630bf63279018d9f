"""`bench.py --gpus N` starts by itself (VERDICT r03 #1a): without WORLD_SIZE it re-executes under torch.distributed.run with N
processes, rank 0 prints ONE JSON line.  Run here as a DRY RUN (VNM_BENCH_DRY_RUN=1: gloo rendezvous on 127.0.0.1, barriers, the
max-over-ranks clock, the JSON contract -- no GPU work; the container has no GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gpus", [2, 3])
def test_bench_self_launch_dry_run(gpus):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VNM_BENCH_DRY_RUN"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                 # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == gpus and j["rccl_ranks"] == gpus and j["steps"] == 3 and j["dry_run"] is True
    assert "stream" in j["config"]["workload"]      # configs[3] is the default workload of a multi-rank run
    # the clock is the MAX over ranks: rank r sleeps (1 + r) ms per step
    assert j["ms_per_step"] >= gpus * 1.0 * 0.9


def test_bench_launched_by_the_driver_dry_run():
    """... and the driver's own launch line still works: torch.distributed.run sets WORLD_SIZE, bench.py does not launch again."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["VNM_BENCH_DRY_RUN"] = "1"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
