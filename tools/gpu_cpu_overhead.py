"""How much host time does one bench step take to ENQUEUE (no GPU sync) vs the GPU time it represents?
Emulates one rank of an 8-rank K/V-sharded job (n_local = n/8) with a one-rank RCCL communicator."""
import importlib, os, sys, time
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
be = pkg.HipBackend(dev)
m, n, d = 32768, 65536 // 8, 128
g = torch.Generator(device=dev); g.manual_seed(1)
Q = torch.rand((m, d), generator=g, device=dev, dtype=torch.float64) * 2 - 1
K = torch.rand((n, d), generator=g, device=dev, dtype=torch.float64) * 2 - 1
V = torch.rand((n, d), generator=g, device=dev, dtype=torch.float64) * 2 - 1
for merge in ("allreduce", "gather"):
    sa = pkg.ShardedAttention(be, 0, 1, dist, force_collectives=True, merge=merge)
    def step():
        sa.load_kv_shard_f64(K, V, n, d, d)
        qf = sa.convert_q(Q)
        c, lm, ls = sa.batch_partial(qf)
        c, w = sa.batch_merge(c, lm, ls, async_reduce=True)
        if w is not None: w.wait()
        return be.cvt_f2d(c, d)
    for _ in range(5): step()
    torch.cuda.synchronize()
    N = 50
    t0 = time.perf_counter()
    for _ in range(N): step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s: host enqueue %.3f ms/step, wall incl. GPU %.3f ms/step" % (merge, t_enq / N * 1e3, t_all / N * 1e3))
dist.destroy_process_group()
