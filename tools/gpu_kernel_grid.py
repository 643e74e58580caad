"""Kernel-only rate of the fused fp32 kernel over (query rows, key rows) launch shapes at d=128,
device level, resident operands -- what the host pipeline's choice of Q batch and K/V chunk sizes
is based on.  One JSON line per shape."""
import importlib, os, sys, json, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")
d = 128
prec = "bf16" if "--bf16" in sys.argv else "f32"
dims = [int(a) for a in sys.argv[1:] if a.isdigit()] or [d]
def warm_clock(fn, ms=60.0):
    """from idle the core clock needs ~20 ms of matrix work to reach its plateau
    (profiles/r02/short_step_clock_ramp.log): warm by time before the first timed shape"""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()


warmed = False
for d in dims:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    Kfull = torch.rand((65536, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    Vfull = torch.rand((65536, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    Qfull = torch.rand((32768, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    for rows in (2048, 4096, 8192, 16384, 32768):
        for keys in (1024, 2048, 4096, 8192, 16384, 65536):
            sa = pkg.ShardedAttention(be, precision=prec)
            sa.load_kv_shard_f64(Kfull[:keys].contiguous(), Vfull[:keys].contiguous(), keys, d, d)
            qf = sa.convert_q(Qfull[:rows].contiguous())
            for _ in range(3):
                sa.batch_partial(qf)
            if not warmed:
                warm_clock(lambda: sa.batch_partial(qf))
                warmed = True
            reps = max(3, min(50, int(2e13 / (4.0 * rows * keys * d) / 100)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                sa.batch_partial(qf)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fn = pkg.load().sdpa_dev_kv_splits_bf16 if prec == "bf16" else pkg.load().sdpa_dev_kv_splits
            print(json.dumps({"prec": prec, "d": d, "rows": rows, "keys": keys, "splits": fn(rows, keys, d, d),
                              "ms": round(ms, 4), "tflops": round(4.0 * rows * keys * d / (ms * 1e-3) / 1e12, 1)}), flush=True)
