"""GPU (-m gpu): bench.py's N > 1 path end to end on the ONE GPU a test box has.

`python bench.py --gpus N` is one command line at any N, like the reference (`mpirun -np P ./attention-mpi
file`, README.md:137-141; attention-mpi.c:503-506): with no WORLD_SIZE in the environment it counts the
GPUs, refuses when there are fewer than N, and otherwise launches its own N ranks under
torch.distributed.run.  The dev mode SDPA_BENCH_BACKEND=gloo SDPA_BENCH_SHARE_GPU=1 puts every rank
on cuda:0 (metric prefixed DRY RUN), so the launch logic, the per-rank seeded shards, the cross-step reduce
pipeline (attention-mpi.c:364-380) and the N > 1 parity re-draw run for real here; every rank's compute
is the HIP path, only the transport under torch.distributed differs from the 8-GPU run."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
BENCH = os.path.join(ROOT, "bench.py")
QUICK = ["--steps", "2", "--warmup", "1", "--prewarm-ms", "0", "--no-cpu-baseline", "--no-boundary"]


def run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


def json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="checks the fewer-GPUs-than-ranks failure")
def test_more_ranks_than_gpus_is_a_hard_failure():
    r = run(["--gpus", "2"] + QUICK)
    assert r.returncode != 0
    assert "2 GPUs requested, 1 visible" in r.stderr
    assert json_lines(r.stdout) == []


DRY = {"SDPA_BENCH_BACKEND": "gloo", "SDPA_BENCH_SHARE_GPU": "1"}


@pytest.mark.parametrize("world,extra", [
    (2, []),                                  # one batch per step, all-gather merge (the default)
    (3, ["--q-batch", "3000"]),               # ragged K/V shards (8192 = 2731 + 2731 + 2730), 3 batches per step
    (2, ["--merge", "allreduce"]),            # the reference's literal all-reduce(MAX) / all-reduce(SUM)
    (2, ["--plan", "qrows"]),                 # query rows sharded, gather of finished rows
])
def test_self_launched_dry_run_world(world, extra):
    r = run(["--gpus", str(world), "--workload", "config2"] + QUICK + extra, env=DRY)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    j = lines[0]
    assert j["metric"].startswith("DRY RUN (gloo") and "not a result" in j["metric"]
    assert j["n_gpus"] == world and j["steps"] == 2 and j["warmup"] == 1
    assert j["rccl"]["world_size"] == world and j["rccl"]["backend"] == "gloo"
    assert j["rccl"]["allreduce_of_ones"] == float(world)
    assert j["parity_max_err"] <= j["parity_tol"]            # rank 0 re-drew every rank's seeded shard
    assert j["cpu_baseline"] is None and j["scaling"] == "strong"
    if "--plan" not in extra:
        assert j["config"]["kv_rows_per_gpu"] == 8192 // world + (1 if 8192 % world else 0)
    assert j["roofline"]["launches"] == 2 * j["config"]["q_batches"]


def test_single_gpu_line_keeps_the_round_2_keys():
    """N = 1 stays what BENCH_r02.json recorded: same keys (plus additions), same metric/config strings."""
    r = run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-boundary"])
    assert r.returncode == 0, r.stderr[-3000:]
    j = json_lines(r.stdout)[0]
    old = json.load(open(os.path.join(ROOT, "BENCH_r02.json")))["parsed"]
    assert set(old) - {"extra_keys"} <= set(j)
    extra = old.get("extra_keys") or {}
    assert set(extra) - {"boundary"} <= set(j)          # the driver files non-contract keys under extra_keys
    assert set(old["roofline"]) <= set(j["roofline"]) and set(old["config"]) <= set(j["config"])
    for k in ("metric", "unit", "n_gpus", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert j[k] == old[k], k
    assert j["config"] == old["config"]
    assert j["rccl"] is None and j["parity_max_err"] <= j["parity_tol"]
