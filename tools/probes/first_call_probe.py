"""Why is the FIRST boundary call of a process 1 ms slower than the fourth?  A fresh process per variant: sdpa_prepare() as shipped, then
(a) the timed call at once, (b) 60 ms of the fused kernel on REAL (random) resident data first, (c) 60 ms of idle first.
    python tools/probes/first_call_probe.py"""
import importlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
    lib = pkg.load()
    m, n, d = 32768, 65536, 128
    rng = np.random.default_rng(0)
    Q, K, V = (rng.uniform(-1, 1, s) for s in ((m, d), (n, d), (n, d)))
    R = np.zeros((m, d))
    pkg.init(1)
    be = pkg.HipBackend("cuda:0")
    sa = pkg.ShardedAttention(be)
    if sys.argv[1].startswith("real"):
        sa.load_kv_shard_f64(torch.from_numpy(K).cuda(), torch.from_numpy(V).cuda(), n, d, d)
        qf = sa.convert_q(torch.from_numpy(Q).cuda())
        torch.cuda.synchronize()
    time.sleep(0.5)
    lib.sdpa_prepare(m, n, d, d, 0)
    if sys.argv[1].startswith("real"):
        for _ in range(int(sys.argv[1][4:] or 8)):
            sa.batch_partial(qf)
        torch.cuda.synchronize()
    elif sys.argv[1] == "idle":
        time.sleep(0.06)
    out = []
    for it in range(4):
        lib.sdpa_attention_f64(Q.ctypes.data, K.ctypes.data, V.ctypes.data, R.ctypes.data, m, n, d, d, 0)
        t = pkg.last_timing()
        out.append("%.0f (kernel %.0f)" % (t["total_us"], t["kernel_us"]))
    print(sys.argv[1], " | ".join(out), flush=True)
    sys.exit(0)
for rep in range(2):
    for v, dbg in (("plain", ""), ("real4", "")):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), v], capture_output=True, text=True, env=dict(os.environ, SDPA_DEBUG=dbg))
        print("%-8s" % v, (r.stdout.strip().split("\n") or ["?"])[-1] if r.returncode == 0 else "FAILED " + r.stderr[-300:], flush=True)
