"""CPU: what bench.py promises before it touches a GPU -- it refuses to run without one (there is no
CPU leg that could stand in for the measured path), refuses a --gpus/WORLD_SIZE mismatch, stamps the
PMC traffic figure with the kernel sources it was measured on, and derives its untimed clock
pre-warm from the shape alone (every rank of an N-rank job must issue the same collectives)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")
HAS_GPU = torch.cuda.is_available()


def run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=300, env=e)


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_bench_without_a_gpu_fails_loudly_and_prints_no_result():
    r = run(["--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_bench_refuses_a_world_size_mismatch():
    r = run(["--gpus", "2"])
    assert r.returncode != 0 and "torch.distributed.run" in r.stderr and r.stdout == ""
    r = run(["--gpus", "1"], env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4 does not match --gpus 1" in r.stderr


def test_traffic_stamp_names_the_shipped_kernel_sources():
    sys.path.insert(0, ROOT)
    import bench
    stamp = bench.kernel_source_stamp()
    assert len(stamp) == 16 and int(stamp, 16) >= 0
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    # the committed PMC figure belongs to the committed kernel sources (else bench prints traffic: null)
    assert tj["kernel_src_sha16"] == stamp
    assert tj["per_launch_bytes"] == pytest.approx(
        tj["fetch_size_kib"] * 1024 * 2 + tj["write_size_kib"] * 1024, rel=1e-9)


def test_clock_prewarm_count_is_a_function_of_the_shape_only():
    sys.path.insert(0, ROOT)
    import bench
    f = bench.prewarm_step_count
    assert f(32768, 65536, 128, "f32", 60.0) == 7          # metric shape, one GPU: 7 x 7.7 ms
    assert f(32768, 8192, 128, "f32", 60.0) == 52          # one rank's share at N = 8
    assert f(8192, 8192, 128, "f32", 60.0) == 190          # config 2
    assert f(512, 512, 64, "f32", 60.0) == 400             # capped
    assert f(32768, 65536, 512, "bf16", 60.0) == 14
    assert f(32768, 65536, 128, "f32", 0.0) == 0           # --prewarm-ms 0: off
