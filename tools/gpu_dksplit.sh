#!/bin/bash
# dk-split fp32 kernel (256 < dk <= 512): parity, then the config-5 shape in fp32 against the
# VALU any-shape kernel it replaces ($SDPA_TUNE=8).
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shapes_device_level or steep" 2>&1 | tail -6
timeout 600 python - <<'PY'
import importlib, os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend(torch.device("cuda", 0))
for (m, n, d) in ((8192, 8192, 512), (32768, 65536, 512), (8192, 16384, 384)):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    Q = torch.rand((m, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    K = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    V = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    sa = pkg.ShardedAttention(be)
    sa.load_kv_shard_f64(K, V, n, d, d)
    qf = sa.convert_q(Q)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c, lm, ls = sa.batch_partial(qf)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("m=%d n=%d d=%d  %.2f ms  %.1f TFLOP/s (%.0f %% of 157.3)" % (m, n, d, dt * 1e3, 4.0 * m * n * d / dt / 1e12, 4.0 * m * n * d / dt / 1.573e12))
PY
