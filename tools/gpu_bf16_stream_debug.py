"""one streamed bf16 call at a given shape (debugging aid: run under AMD_LOG_LEVEL=4 to see which engine carries each copy)
    python tools/gpu_bf16_stream_debug.py m n dk dv"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
m, n, dk, dv = (int(x) for x in sys.argv[1:5])
rng = np.random.default_rng(0)
Q, K, V = (rng.uniform(-1, 1, s) for s in ((m, dk), (n, dk), (n, dv)))
pkg.init(1)
print("plan", pkg.plan(m, n, dk, dv, 2, 1)["r"][0]["stream"], flush=True)
sys.stderr.write("==== CALL BEGINS\n"); sys.stderr.flush()
try:
    out = pkg.attention(Q, K, V, precision="bf16")
    print("ok", pkg.last_timing()["streamed"], pkg.last_timing()["last_kernel"], float(np.abs(out).max()))
except Exception as e:  # noqa: BLE001
    print("FAILED", e)
