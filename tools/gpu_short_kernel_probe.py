"""Why does a SHORT fused launch take 10-15 % longer inside a bench step than back to back
(config 2: 0.309 vs 0.268 ms; one rank's share at N = 8: 1.095 vs 0.994 ms; the 7.67 ms metric
launch is unaffected)?  Runs the fused fp32 kernel at (rows, keys) in four modes, 30 launches each,
in this order, so that a rocprofv3 --kernel-trace of the process can be cut by launch index:
  0  fused back to back                       1  fused, a HIP event pair around every launch
  2  K/V/Q converts + fused + finish per iteration (a bench step), no events      3  = 2 + events
Prints the event-derived and wall times per mode; the trace gives the kernel's own duration."""
import importlib, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")
d, N = 128, 30
shapes = [(int(a.split("x")[0]), int(a.split("x")[1])) for a in sys.argv[1:]] or [(8192, 8192), (32768, 8192)]
for rows, keys in shapes:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    K64 = torch.rand((keys, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    V64 = torch.rand((keys, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    Q64 = torch.rand((rows, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    sa = pkg.ShardedAttention(be)
    sa.load_kv_shard_f64(K64, V64, keys, d, d)
    qf = sa.convert_q(Q64)
    for _ in range(5):
        sa.batch_partial(qf)
    for mode in range(4):
        evs = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            if mode >= 2:
                sa.load_kv_shard_f64(K64, V64, keys, d, d)
                qf = sa.convert_q(Q64)
            if mode & 1:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            contrib, lmax, lsum = sa.batch_partial(qf)
            if mode & 1:
                e1.record()
                evs.append((e0, e1))
            if mode >= 2:
                be.finish_f64(contrib, lsum, d)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / N * 1e3
        ev = sum(a.elapsed_time(b) for a, b in evs) / len(evs) if evs else None
        print(json.dumps({"rows": rows, "keys": keys, "mode": mode, "wall_ms_per_iter": round(wall, 4),
                          "event_ms": None if ev is None else round(ev, 4)}), flush=True)
