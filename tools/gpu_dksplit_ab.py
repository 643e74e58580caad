"""A/B at 128 < dk <= 256: fused_partial_kernel (dv in 128-column chunks, score tile recomputed per
chunk) vs the dk-split kernel ($SDPA_TUNE=128).  Run once per setting: the library reads the
variable once."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend(torch.device("cuda", 0))
for (m, n, dk, dv) in ((8192, 16384, 256, 256), (32768, 65536, 256, 256), (8192, 16384, 192, 192), (8192, 16384, 256, 64),
                      (8192, 16384, 160, 512)):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    Q = torch.rand((m, dk), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    K = torch.rand((n, dk), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    V = torch.rand((n, dv), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    sa = pkg.ShardedAttention(be)
    sa.load_kv_shard_f64(K, V, n, dk, dv)
    qf = sa.convert_q(Q)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c, lm, ls = sa.batch_partial(qf)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fl = 2.0 * m * n * (dk + dv)
    print("TUNE=%s m=%d n=%d dk=%d dv=%d  %.2f ms  %.1f TFLOP/s" % (os.environ.get("SDPA_TUNE", "0"), m, n, dk, dv, dt * 1e3, fl / dt / 1e12))
