#!/bin/bash
# round 5, call 3: why did the streamed launch's ready words never arrive in the engine (call 2) when the first probe saw them?
# The probe again with a kernel that owns every register of every SIMD, the engine's streams and events; then the engine
# itself with a short timeout, saying WHICH word timed out, with the default and with more hardware queues.
O=gpurun_out/r05_03; mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/probes/stream_flag_probe2 > $O/stream_flag_probe2.log 2>&1; echo "probe2 rc=$?"; cut -c1-330 $O/stream_flag_probe2.log
cat > /tmp/diag.py <<'P'
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
rng = np.random.default_rng(0)
for (m, n, d) in ((8192, 8192, 128), (32768, 16384, 128)):
    Q, K, V = (rng.uniform(-1, 1, s) for s in ((m, d), (n, d), (n, d)))
    pkg.init(1)
    for rep in range(2):
        t0 = time.perf_counter()
        try:
            pkg.attention(Q, K, V)
            t = pkg.last_timing()
            print((m, n, d), "ok", "streamed", t["streamed"], "total_ms", round(t["total_us"] / 1e3, 3), flush=True)
        except Exception as e:
            print((m, n, d), "FAILED after %.2f s: %s" % (time.perf_counter() - t0, e), flush=True)
P
for envs in "SDPA_NOP=1" "GPU_MAX_HW_QUEUES=8" "SDPA_STREAM_CHUNK_MIN=1024"; do
  echo "== $envs"; env SDPA_STREAM_TIMEOUT_MS=300 $envs timeout 120 python /tmp/diag.py 2>&1 | grep -v "^$" | cut -c1-300 | tail -12
done
