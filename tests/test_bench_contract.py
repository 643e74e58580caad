"""CPU: what bench.py promises before it touches a GPU -- it refuses to run without one (there is no
CPU leg that could stand in for the measured path), refuses a --gpus/WORLD_SIZE mismatch, launches its own ranks for --gpus N (after counting GPUs), stamps the
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
    r = run(["--gpus", "1"], env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4 does not match --gpus 1" in r.stderr


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_bench_gpus_n_without_a_launcher_launches_itself_and_needs_gpus():
    """`python bench.py --gpus N` is one command line at any N (the reference: README.md:137-141): with no
    WORLD_SIZE it starts its own ranks -- after checking that the GPUs exist.  Here there are none."""
    r = run(["--gpus", "2"])
    assert r.returncode != 0 and "needs a GPU" in r.stderr and r.stdout == ""


def test_self_launch_uses_the_drivers_launch_line():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(8, 29611, ["--gpus", "8", "--steps", "5", "--warmup", "2"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    assert cmd[-7] == BENCH and cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]


def test_host_staged_gloo_adapter_world_2(tmp_path):
    """the dev-mode collective adapter (SDPA_BENCH_BACKEND=gloo) on CPU tensors, world size 2:
    all-gather layout [world, 2, m] as engine.batch_merge reads it, reduce to the root, all-reduce MAX"""
    child = r'''
import sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import bench
rank = int(sys.argv[2])
dist.init_process_group("gloo", init_method="file://" + sys.argv[3], rank=rank, world_size=2)
d = bench.HostStagedDist(dist)
mine = torch.stack((torch.full((5,), float(rank)), torch.full((5,), 10.0 + rank)))
stats = torch.empty((2, 2, 5))
d.all_gather_into_tensor(stats.view(-1, 5), mine)
assert stats[0, 0, 0] == 0 and stats[0, 1, 0] == 10 and stats[1, 0, 0] == 1 and stats[1, 1, 0] == 11
t = torch.full((3,), 1.0 + rank)
assert d.reduce(t, dst=0, op=d.ReduceOp.SUM, async_op=True) is None
assert rank != 0 or bool((t == 3).all())
x = torch.tensor([float(rank)])
d.all_reduce(x, op=d.ReduceOp.MAX)
assert x.item() == 1.0 and d.get_world_size() == 2
d.barrier()
d.destroy_process_group()
'''
    store = str(tmp_path / "store")
    ps = [subprocess.Popen([sys.executable, "-c", child, ROOT, str(r), store], stderr=subprocess.PIPE, text=True)
          for r in range(2)]
    for p in ps:
        _, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-1500:]


def test_traffic_stamp_names_the_shipped_kernel_sources():
    sys.path.insert(0, ROOT)
    import bench
    stamp = bench.kernel_source_stamp()
    assert len(stamp) == 16 and int(stamp, 16) >= 0 and bench.kernel_source_stamp("bf16") != stamp
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    entries = tj.get("entries") or {"headline/f32": tj}
    for key, e in entries.items():
        assert e["per_launch_bytes"] == pytest.approx(e["fetch_size_kib"] * 1024 * 2 + e["write_size_kib"] * 1024, rel=1e-9)
        if "hbm_gbps" in e:
            assert e["hbm_gbps"] == pytest.approx(e["per_launch_bytes"] / (e["steady_avg_ms"] * 1e-3) / 1e9, rel=1e-6)
            assert 0.0 < e["mfma_util"] <= 1.0
    # a PMC figure is only quoted for the kernel sources it was measured on (else bench prints null)
    head = entries.get("headline/f32")
    got = bench.pmc_stamp("headline", "f32")
    if head is None or head["kernel_src_sha16"] != stamp:
        assert got == {"traffic": None, "hbm_gbps": None, "mfma_util": None, "provenance": None}
        pytest.skip("profiles/traffic_latest.json was measured on older kernel sources: bench.py prints "
                    "traffic: null until tools/gpu_profile.sh has been re-run")
    assert got["traffic"] == head["per_launch_bytes"]
    # the figures are copied from a committed profile, and the line says so next to them
    assert got["provenance"]["file"] == "profiles/traffic_latest.json" and got["provenance"]["kernel_src_sha16"] == stamp
    assert bench.pmc_stamp("config1", "f32")["traffic"] is None          # never profiled: never quoted


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


def test_abort_trace_names_the_native_thread_that_aborts(tmp_path):
    """tests/abort_trace.c (loaded by conftest.py under $SDPA_ABORT_TRACE, set by tools/gpu_flaky_hunt.sh): when
    some native thread of a -m gpu run calls abort(), the log must say WHICH thread and from which module --
    faulthandler alone shows the Python main thread, wherever it happened to be."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no gcc here")
    so = str(tmp_path / "abort_trace.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "abort_trace.c")])
    code = ("import ctypes, threading\n"
            "lib = ctypes.CDLL(%r); assert lib.abort_trace_install(2) == 0\n"
            "libc = ctypes.CDLL(None)\n"
            "t = threading.Thread(target=lambda: libc.abort(), name='doomed'); t.start(); t.join()\n" % so)
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == -6, r
    assert "SIGABRT raised on thread" in r.stderr and "abort+0x" in r.stderr and "end of native backtrace" in r.stderr, r.stderr
    assert r.stderr.index("SIGABRT raised") < r.stderr.index("Fatal Python error"), r.stderr      # then faulthandler's dump


def test_other_configs_record_names_the_baseline_configs_and_the_n1_reference_is_committed():
    """bench.py at N = 1 measures BASELINE configs 2, 4, 5 (bf16 and fp32) beside the headline (tests/test_gpu_bench_line.py
    runs it); an N > 1 line states its speedup against the committed N = 1 figures"""
    sys.path.insert(0, ROOT)
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    names = {k: (bench.WORKLOADS[wl], prec) for k, wl, prec in bench.OTHER_CONFIGS}
    assert sorted(names) == ["config2", "config4", "config5_bf16", "config5_f32"]
    assert names["config2"][0] == dict(m=8192, n=8192, d=128) and "m=8192 n=8192" in base["configs"][1]
    assert names["config4"][0] == dict(m=131072, n=65536, d=128) and "m=131072 n=65536" in base["configs"][3]
    assert names["config5_bf16"] == (dict(m=32768, n=65536, d=512), "bf16") and "d_k=d_v=512" in base["configs"][4]
    ref = bench.n1_reference()
    for wl in ("headline", "config3"):
        assert ref[wl]["ms_per_step"] > 0 and ref[wl]["latency_ms"] > 0 and 0.5 < ref[wl]["frac"] < 1.0
