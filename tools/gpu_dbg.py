"""debug aid: max error of the bf16 device-level path on a few shapes, several repeats"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")
orc = O.Oracle()
for (m, n, dk, dv, dist) in [(130, 333, 128, 128, "D2"), (257, 2048, 128, 128, "D3"), (700, 5128, 128, 128, "D2"),
                             (700, 5064, 64, 128, "D2"), (700, 5128, 128, 64, "D2"), (700, 5064, 64, 64, "D2"),
                             (700, 5256, 256, 256, "D2")]:
    Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=m + n + 1)
    want = orc.attention_f64(Q, K, V)
    sa = pkg.ShardedAttention(be, precision="bf16")
    sa.load_kv_from_root(K, V, n, dk, dv)
    qb = sa.convert_q(torch.from_numpy(Q).cuda())
    errs = []
    first = None
    for it in range(6):
        contrib, lmax, lsum = sa.batch_partial(qb)
        got = be.finish_f64(contrib, lsum, dv).cpu().numpy()
        errs.append(float(np.abs(got - want).max()))
        if first is None: first = got
        nd = int((got != first).sum())
    print((m, n, dk, dv, dist), "tol %.3g" % (1e-2 * max(1, np.abs(V).max())), "errs", ["%.3g" % e for e in errs],
          "differing elements vs run 0:", nd, "splits", pkg.load().sdpa_dev_kv_splits_bf16(m, n, dk, dv), flush=True)
