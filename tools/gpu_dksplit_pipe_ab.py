"""fp32 dk-split kernel: the software-pipelined form against the serial-phase one ($SDPA_DKSPLIT_PIPE=1/0,
read per launch) -- bit for bit over head dims, ragged shard lengths, one- and two-tile shards, in-GPU K/V
splits, ragged query-row counts; then the kernel-only rate of both at m=32768, n=65536."""
import importlib, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")


def run(m, n, dk, dv, pipe, seed=1, scale=1.0):
    os.environ["SDPA_DKSPLIT_PIPE"] = "1" if pipe else "0"
    pkg.reload_env()
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    Q = (torch.rand((m, dk), generator=g, device="cuda", dtype=torch.float64) * 2 - 1) * scale
    K = (torch.rand((n, dk), generator=g, device="cuda", dtype=torch.float64) * 2 - 1) * scale
    V = torch.rand((n, dv), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    sa = pkg.ShardedAttention(be)
    sa.load_kv_shard_f64(K, V, n, dk, dv)
    qf = sa.convert_q(Q)
    out = sa.batch_partial(qf)
    torch.cuda.synchronize()
    return [t.clone() for t in out], (sa, qf)


bad = 0
cases = []
for d in (512, 384, 320, 264, 768, 1024, 640):
    for (m, n) in ((64, 32), (64, 33), (100, 64), (130, 95), (257, 1000), (2048, 4096), (1000, 8191), (33, 1)):
        cases.append((m, n, d, d))
for (dk, dv) in ((512, 64), (512, 200), (300, 512), (200, 256), (160, 384), (1024, 32), (700, 130)):
    for (m, n) in ((96, 777), (1024, 2049)):
        cases.append((m, n, dk, dv))
cases.append((8192, 65536, 512, 512))          # BASELINE config 5's dims, in-GPU splits
for i, (m, n, dk, dv) in enumerate(cases):
    a, _ = run(m, n, dk, dv, True, seed=i, scale=3.0 if i % 3 == 0 else 1.0)
    b, _ = run(m, n, dk, dv, False, seed=i, scale=3.0 if i % 3 == 0 else 1.0)
    a[0], b[0] = a[0][:, :dv], b[0][:, :dv]          # columns past dv are padding nobody writes
    same = all(torch.equal(x, y) for x, y in zip(a, b))
    finite = all(bool(torch.isfinite(x).all()) for x in a[:1])
    if not (same and finite):
        bad += 1
        diff = max(float((x.double() - y.double()).abs().max()) for x, y in zip(a, b))
        print("MISMATCH", (m, n, dk, dv), "max|diff|", diff, "finite", finite, flush=True)
        for x, y in zip(a, b):
            ne = (x != y) | (x.isnan() != y.isnan())
            if ne.any():
                idx = ne.nonzero()
                print("   tensor", tuple(x.shape), "differing", int(ne.sum()), "first", idx[:3].tolist(), "last", idx[-3:].tolist(),
                      "cols", sorted(set(idx[:, -1].tolist()))[:8] if idx.shape[1] > 1 else "", flush=True)
print("bitwise: %d cases, %d mismatching" % (len(cases), bad), flush=True)

for d in [int(x) for x in sys.argv[1:]] or [512, 384, 768, 1024]:
    m, n = 32768, 65536
    for pipe in (0, 1, 0, 1):
        _, (sa, qf) = run(m, n, d, d, pipe)
        for _ in range(2): sa.batch_partial(qf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 4
        e0.record()
        for _ in range(reps): sa.batch_partial(qf)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(json.dumps({"d": d, "pipelined": pipe, "kernel_ms": round(ms, 3), "tflops": round(4.0 * m * n * d / ms / 1e9, 1),
                          "frac_of_157.3": round(4.0 * m * n * d / ms / 1e9 / 157.3, 3)}), flush=True)
        del sa, qf
