"""fp32 kernel rate at head dims above 128 (register-staged MFMA kernel with dv chunks)."""
import importlib, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")
for (m, n, dk, dv) in [(16384, 16384, 256, 256), (16384, 16384, 192, 192), (16384, 16384, 128, 256), (4096, 8192, 512, 512)]:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    Q = torch.rand((m, dk), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    K = torch.rand((n, dk), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    V = torch.rand((n, dv), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    sa = pkg.ShardedAttention(be)
    sa.load_kv_shard_f64(K, V, n, dk, dv)
    qf = sa.convert_q(Q)
    for _ in range(2): sa.batch_partial(qf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps): sa.batch_partial(qf)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"shape": [m, n, dk, dv], "kernel_ms": round(ms, 3), "tflops": round(2.0 * m * n * (dk + dv) / ms / 1e9, 1)}))
