"""GPU (-m gpu): the ONE line `python bench.py` prints at N = 1 carries what the driver's single command must yield
(VERDICT r4 item 1): the headline with the kernel the LAUNCHER recorded, the un-pipelined latency next to the per-step
time, the config-3 scaling record, and a `configs` record with BASELINE configs 2, 4 and 5 (bf16, and its dims in fp32)
-- each with the fused kernel's time and fraction of its peak, the boundary (host fp64 in/out) and a parity figure."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_at_n1_carries_every_baseline_config():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--min-gpu-seconds", "0", "--prewarm-ms", "20"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    j = json.loads(lines[0])
    # the contract keys of round 4 are all still there
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "boundary", "scaling_config3", "parity_max_err"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["dtype"] == "f32" and j["vs_baseline"] is None and j["parity_max_err"] <= j["parity_tol"]
    roof = j["roofline"]
    assert roof["bound"] == "mfma" and roof["peak"] == 157.3 and 0.5 < roof["frac"] <= 1.0
    # what launched, as the launcher recorded it (not a guess from the shape)
    assert roof["kernel"] == "sdpa::fused_pipelined_kernel<128,128,0,0>", roof["kernel"]
    assert roof["kernel_launch"]["grid"] == 512 and roof["kernel_launch"]["splits"] == 2 and roof["kernel_launch"]["stream_k"] == 0
    assert "sdpa_dev_last_launch" in roof["kernel_launch"]["source"]
    assert 0.9 * j["ms_per_step"] <= j["latency_ms"] <= 1.5 * j["ms_per_step"]
    s3 = j["scaling_config3"]
    assert "error" not in s3 and s3["latency_ms"] > 0 and s3["boundary_ms"] > s3["kernel_ms_avg"] and s3["kernel"] == roof["kernel"]
    cfg = j["configs"]
    assert sorted(cfg) == ["config2", "config4", "config5_bf16", "config5_f32"]
    want_kernel = {"config2": "fused_pipelined_kernel<128,128,0,0>", "config4": "fused_pipelined_kernel<128,128,0,0>",
                   "config5_bf16": "fused_bf16_tandem_kernel<512>", "config5_f32": "fused_dksplit_pipe_kernel<128,128,2>"}
    for k, rec in cfg.items():
        assert "error" not in rec, (k, rec.get("error"))
        assert rec["kernel"] == "sdpa::" + want_kernel[k], (k, rec["kernel"])
        assert rec["kernel_ms_avg"] > 0 and 0.2 < rec["frac"] <= 1.0 and rec["parity_max_err"] <= rec["parity_tol"], (k, rec)
        assert rec["boundary_ms"] > rec["kernel_ms_avg"] and rec["boundary"]["parity_max_err"] <= rec["boundary"]["parity_tol"], k
        assert rec["peak_tflops"] == (2500.0 if k == "config5_bf16" else 157.3)
    # the boundary call of the fp32 BASELINE shapes with d <= 128 is one streamed launch (round 5)
    assert j["boundary"]["streamed"] == 1 and j["boundary"]["fused_launches"] == 1
    assert j["boundary"]["last_kernel"] == "sdpa::fused_pipelined_stream_kernel<128,128>"
    assert cfg["config2"]["boundary"]["streamed"] == 1 and cfg["config5_f32"]["boundary"]["streamed"] == 0
    # round 6: the cold one-shot CLI runs (the reference's literal use, attention.c:179-189) and the host feed model ride along
    cli = j["cli_one_shot"]
    for k in ("headline", "config2"):
        assert "error" not in cli[k], (k, cli[k])
        assert cli[k]["correct"] is True and len(cli[k]["elapsed_ms"]) == 3 and cli[k]["min_ms"] <= cli[k]["median_ms"], cli[k]
    assert cli["headline"]["median_ms"] < 3.0 * j["boundary"]["ms"] and "head" in (cli["headline"]["stages_last_run"] or "")
    fm = j["feed_model"]
    assert fm["pageable"] in ("host", "device") and fm["t_host_ms"] > 0 and fm["t_kernel_ms"] > fm["t_link_ms"] > 0, fm
