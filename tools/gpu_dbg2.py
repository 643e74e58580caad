"""debug aid: where the bf16 duo kernel's output differs from the general kernel's (same inputs)"""
import importlib, os, sys, subprocess, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
be = pkg.HipBackend("cuda:0")
orc = O.Oracle()
m, n, dk, dv, dist = 257, 2048, 128, 128, "D3"
Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=m + n + 1)
want = orc.attention_f64(Q, K, V)
sa = pkg.ShardedAttention(be, precision="bf16")
sa.load_kv_from_root(K, V, n, dk, dv)
qb = sa.convert_q(torch.from_numpy(Q).cuda())
for it in range(3):
    contrib, lmax, lsum = sa.batch_partial(qb)
    got = be.finish_f64(contrib, lsum, dv).cpu().numpy()
    err = np.abs(got - want).max(axis=1)
    bad = np.nonzero(err > 0.1)[0]
    print("run", it, "bad rows:", len(bad), "first", bad[:40].tolist(), "lmax nan/inf:", int((~torch.isfinite(lmax)).sum()),
          "lsum<=0:", int((lsum <= 0).sum()), flush=True)
    if len(bad):
        r = int(bad[0])
        print("  row", r, "err", float(err[r]), "lmax", float(lmax[r]), "lsum", float(lsum[r]),
              "true max score", float(((Q[r] @ K.T) / np.sqrt(dk)).max()))
        cols = np.nonzero(np.abs(got[r] - want[r]) > 0.1)[0]
        print("  bad cols in that row:", len(cols), cols[:32].tolist())
