"""The cold one-shot CLI against sdpa_prepare()'s clock warm-up: bench.py's cli_one_shot (3 fresh processes per setting) at the metric shape under
several $SDPA_PREPARE_WARM_MS.    python tools/probes/cli_cold_warmup.py [ms ...]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import torch
for ms in (sys.argv[1:] or ["0", "60", "200", "500"]):
    ms, _, dbg = ms.partition(":")            # "60:prepare_zero=1" = 60 ms with $SDPA_DEBUG=prepare_zero=1
    os.environ["SDPA_PREPARE_WARM_MS"] = ms
    os.environ["SDPA_DEBUG"] = dbg
    r = bench.cli_one_shot(torch.device("cuda:0"), names=("headline",), runs=3)["headline"]
    print("warm_ms", ms, dbg, json.dumps({k: r.get(k) for k in ("elapsed_ms", "stages_last_run", "error")}), flush=True)
