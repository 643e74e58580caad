"""Fused-kernel rate at query-row counts that do NOT fill whole rounds of workgroups (tail quantisation):
kernel ms, in-GPU K/V splits chosen, TFLOP/s.   python tools/gpu_odd_rows.py [--bf16]
Run with SDPA_HIP_LIB=.../lib/variants/libsdpa_hip_r02.so for the round-2 split choice."""
import importlib, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")
prec = "bf16" if "--bf16" in sys.argv else "f32"
shapes = [(32768, 65536, 128), (33000, 65536, 128), (40000, 65536, 128), (48000, 65536, 128), (70000, 65536, 128),
          (20000, 8192, 128), (40000, 65536, 256)] if prec == "f32" else \
         [(32768, 65536, 512), (40000, 65536, 512), (40000, 65536, 128), (70000, 4096, 64)]
lib = pkg.load()
g = torch.Generator(device="cuda"); g.manual_seed(1)
for m, n, d in shapes:
    K = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    V = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    Q = torch.rand((m, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    sa = pkg.ShardedAttention(be, precision=prec)
    sa.load_kv_shard_f64(K, V, n, d, d)
    qf = sa.convert_q(Q)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.08:          # clock pre-warm
        sa.batch_partial(qf)
        torch.cuda.synchronize()
    reps = 8
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sa.batch_partial(qf)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    sp = (lib.sdpa_dev_kv_splits_bf16 if prec == "bf16" else lib.sdpa_dev_kv_splits)(m, n, d, d)
    print(json.dumps({"prec": prec, "shape": [m, n, d], "kv_splits": sp, "kernel_plus_merge_ms": round(ms, 4),
                      "tflops": round(4.0 * m * n * d / (ms * 1e-3) / 1e12, 1)}), flush=True)
