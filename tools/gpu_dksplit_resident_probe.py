"""Is the fp32 dk-split kernel waiting on memory?  Same flop count, K/V of shrinking size (down to what one
XCD's 4 MB L2 holds), query rows grown to compensate: the rate against n."""
import importlib, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")
d = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for n in (65536, 16384, 4096, 2048, 1024):
    m = 32768 * 65536 // n // 4
    m = min(m, 262144)
    for pipe in (0, 1):
        os.environ["SDPA_DKSPLIT_PIPE"] = str(pipe)
        pkg.reload_env()
        os.environ["SDPA_KV_SPLITS"] = "1"
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
        e0.record()
        for _ in range(3): sa.batch_partial(qf)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(json.dumps({"d": d, "m": m, "n": n, "kv_mb": round(2 * n * d * 4 / 1e6, 1), "pipelined": pipe, "kernel_ms": round(ms, 3),
                          "tflops": round(4.0 * m * n * d / ms / 1e9, 1)}), flush=True)
        del sa, qf
