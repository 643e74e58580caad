"""What the bf16 config-5 kernel costs BEHIND other work (round 6): the same resident images, the kernel timed by HIP events, with -- in front of
every launch -- nothing (back to back), the step's converts, 768 MB of memset, 0.3 ms of idle, or one convert alone.  $SDPA_HIP_LIB selects the
library (tools/build_variant.sh nont sdpa_fwd_bf16.hip -DSDPA_CVT_NT=0 = the converters with plain source loads).
    python tools/probes/kernel_behind_converts.py            -> profiles/r06/bf16_kernel_behind_converts.log"""
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")
m, n, d = 32768, 65536, 512
g = torch.Generator(device="cuda"); g.manual_seed(1)
Q = (torch.rand((m, d), dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
K = (torch.rand((n, d), dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
V = (torch.rand((n, d), dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
sa = pkg.ShardedAttention(be, precision="bf16")
sa.load_kv_shard_f64(K, V, n, d, d)
qf = sa.convert_q(Q)
big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
def run(mode, reps=24):
    ts = []
    for i in range(reps):
        if mode == "converts":
            sa.load_kv_shard_f64(K, V, n, d, d); q = sa.convert_q(Q)
        elif mode == "memset":
            big.zero_(); big.zero_(); big.zero_()
        elif mode == "idle":
            torch.cuda.synchronize(); time.sleep(0.0003)
        elif mode == "kconv":
            sa.Kf = be.cvt_d2bf_k(K, d)
        elif mode == "vconv":
            sa.Vf = be.cvt_d2bf_t(V)
        elif mode == "qconv":
            q = sa.convert_q(Q)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sa.batch_partial(qf); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = [a.elapsed_time(b) for a, b in ts][8:]
    print("%-10s kernel ms: median %.4f min %.4f max %.4f" % (mode, float(np.median(v)), min(v), max(v)), flush=True)
import os
print("lib", os.environ.get("SDPA_HIP_LIB", "shipped"), flush=True)
for _ in range(1):
    for mode in ("back2back", "converts", "memset", "idle", "kconv", "vconv", "qconv", "back2back"):
        run(mode)
