"""Back-to-back timing of the fp32 fused kernel alone for several K/V shard lengths (m fixed)."""
import importlib, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")
m, d = 32768, 128
g = torch.Generator(device="cuda"); g.manual_seed(1)
Q = torch.rand((m, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
for n in [65536, 32768, 16384, 8192, 4096]:
    K = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    V = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    sa = pkg.ShardedAttention(be)
    sa.load_kv_shard_f64(K, V, n, d, d)
    qf = sa.convert_q(Q)
    for _ in range(3): sa.batch_partial(qf)
    torch.cuda.synchronize()
    reps = max(5, int(65536 / n) * 5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): sa.batch_partial(qf)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"n_local": n, "kernel_ms": round(ms, 4), "tflops": round(4.0 * m * n * d / ms / 1e9, 1),
                      "kv_splits": pkg.load().sdpa_dev_kv_splits(m, n, d, d)}))
