"""The host feed at P = 8, measured on ONE GPU: the converter pool does not care that the ranks are loopback
ranks -- with SDPA_VIRTUAL_GPUS=8 it converts exactly what eight real ranks would ask it for, in the same order.  For config 3, the
metric shape and config 4, pageable and page-locked caller arrays: the plan's feed model (sdpa_plan_describe: t_host / t_link / t_kernel
and its choice), then the call with the model's choice, with host converts forced and with device converts forced; the pool's own
wall time per call comes from $SDPA_DEBUG=host_cvt_trace=1 (stderr).  The kernels of eight loopback ranks share one chip and one PCIe link:
total_ms is NOT a P = 8 prediction, the pool's "last item done" is.
    python tools/gpu_feed_model.py            -> profiles/r06/feed_model_p8.log"""
import ctypes
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"config3": (32768, 262144, 128), "metric": (32768, 65536, 128), "config4": (131072, 65536, 128)}

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    name, pinned = sys.argv[2], sys.argv[3] == "pinned"
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
    lib = pkg.load()
    m, n, d = SHAPES[name]
    rng = np.random.default_rng(0)

    def buf(shape, fill=True):
        a = rng.uniform(-1, 1, shape) if fill else np.zeros(shape)
        if not pinned:
            return a
        ptr = lib.sdpa_host_alloc(a.nbytes)
        v = np.ctypeslib.as_array((ctypes.c_double * a.size).from_address(ptr)).reshape(a.shape)
        v[...] = a
        return v
    Q, K, V, R = buf((m, d)), buf((n, d)), buf((n, d)), buf((m, d), False)
    pkg.init(1)
    assert lib.sdpa_prepare(m, n, d, d, 0) == 0
    best = None
    for it in range(4):
        sys.stderr.write("call %d\n" % it)
        sys.stderr.flush()
        assert lib.sdpa_attention_f64(Q.ctypes.data, K.ctypes.data, V.ctypes.data, R.ctypes.data, m, n, d, d, 0) == 0
        t = pkg.last_timing()
        if it and (best is None or t["total_us"] < best["total_us"]):
            best = t
    rows = np.arange(0, m, m // 16)
    s = (Q[rows] @ K.T) / np.sqrt(np.float32(d))
    p = np.exp(s - s.max(axis=1, keepdims=True))
    err = float(np.abs(R[rows] - (p / p.sum(axis=1, keepdims=True)) @ V).max())
    print(json.dumps({"total_ms": round(best["total_us"] / 1e3, 3), "head_ms": round(best["head_us"] / 1e3, 3),
                      "kv_stage_ms": round(best["kv_stage_us"] / 1e3, 3), "kernel_ms_rank0": round(best["kernel_us"] / 1e3, 3),
                      "ranks": best["n_gpus"], "host_convert_threads": best["host_convert_threads"], "streamed": best["streamed"],
                      "max_err_16_rows": err}))
    sys.exit(0)

sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
print("# host: %d CPUs online; cgroup cpu.max: %s" % (os.cpu_count(), open("/sys/fs/cgroup/cpu.max").read().strip()
                                                      if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a"), flush=True)
for name, (m, n, d) in SHAPES.items():
    for P in (1, 2, 8):
        print(json.dumps({"shape": name, "plan_ranks": P, "feed": pkg.plan(m, n, d, d, 0, P)["feed"]}), flush=True)
    for mem in ("pageable", "pinned"):
        for tag, env in (("model's choice", {}), ("host converts forced", {"SDPA_HOST_CVT": "1"}), ("device converts forced", {"SDPA_HOST_CVT": "0"})):
            e = dict(os.environ, SDPA_VIRTUAL_GPUS="8", SDPA_DEBUG="host_cvt_trace=1", **env)
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, mem], capture_output=True, text=True, env=e)
            traces = [l for l in r.stderr.split("\n") if "hostcvt trace" in l]
            out = r.stdout.strip().split("\n")[-1] if r.returncode == 0 else "FAILED rc=%d %s" % (r.returncode, r.stderr[-400:])
            print(json.dumps({"shape": name, "arrays": mem, "converts": tag, "virtual_ranks": 8, "result": out,
                              "pool_trace_last_call": traces[-1] if traces else None, "wall_s": round(time.perf_counter() - t0, 1)}), flush=True)
