"""Kernel-only rate of the fp32 path over head dims at m=32768, n=65536 (device level, resident operands)."""
import importlib, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")
for d in [int(x) for x in sys.argv[1:]] or [128, 256, 384, 512]:
    m, n = 32768, 65536
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    Q = torch.rand((m, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    K = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    V = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    sa = pkg.ShardedAttention(be)
    sa.load_kv_shard_f64(K, V, n, d, d)
    qf = sa.convert_q(Q)
    del Q, K, V
    for _ in range(2): sa.batch_partial(qf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 4
    e0.record()
    for _ in range(reps): sa.batch_partial(qf)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"d": d, "kernel_ms": round(ms, 3), "tflops": round(4.0 * m * n * d / ms / 1e9, 1),
                      "frac_of_157.3": round(4.0 * m * n * d / ms / 1e9 / 157.3, 3)}), flush=True)
