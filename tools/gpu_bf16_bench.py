"""Kernel-level timing of the bf16 path at BASELINE config 5 (m=32768 n=65536 d=512) and d=128."""
import importlib, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")
dims = [int(x) for x in sys.argv[1:]] or [512, 384, 256, 128, 64]
for (m, n, d) in [(32768, 65536, d) for d in dims] + [(8192, 8192, 128)]:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    Q = torch.rand((m, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    K = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    V = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    sa = pkg.ShardedAttention(be, precision="bf16")
    sa.load_kv_shard_f64(K, V, n, d, d)
    qb = sa.convert_q(Q)
    del Q, K, V
    # warm by time, not by count: from idle the core clock needs ~20 ms of matrix work to reach its
    # plateau (profiles/r02/short_step_clock_ramp.log); WARM_MS=0 restores the old two-launch warmup
    import time
    warm_s = float(os.environ.get("WARM_MS", "60")) * 1e-3
    for _ in range(2): sa.batch_partial(qb)
    torch.cuda.synchronize()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < warm_s:
        for _ in range(4): sa.batch_partial(qb)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): out = sa.batch_partial(qb)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flop = 4.0 * m * n * d
    print(json.dumps({"shape": [m, n, d], "kernel_ms": ms, "tflops": flop / ms / 1e9,
                      "frac_of_2.5PF": flop / ms / 1e9 / 2500.0,
                      "kv_splits": pkg.load().sdpa_dev_kv_splits_bf16(m, n, d, d),
                      "debug": os.environ.get("SDPA_DEBUG", "")}), flush=True)
