"""GPU (-m gpu): the Python host's collective call path over backend "nccl" (= RCCL) on the one GPU a
test box has -- a one-rank process group, collectives forced on (ShardedAttention(force_collectives=
True)): all_gather_into_tensor + merge_gathered, or the reference's all_reduce(MAX) / all_reduce(SUM),
then the asynchronous reduce(SUM) left in flight under the next Q batch's kernel
(attention-mpi.c:340-380).  World sizes 2 and 3 of the same choreography run over gloo on CPU
(tests/test_sharded_gloo.py); RCCL with more than one rank needs a multi-GPU node.

The group lives in a child process: a hung rendezvous or collective is then a test failure after a
timeout, not a hung suite."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import importlib, os, sys
import numpy as np, torch, torch.distributed as dist
ROOT, PKG, merge, store, out = sys.argv[1:6]
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle as O
pkg = importlib.import_module(PKG)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1, device_id=dev)
try:
    m, n, dk, dv, B = 300, 1000, 128, 128, 128            # 3 Q batches, ragged last one
    Q, K, V = O.make_inputs(m, n, dk, dv, "D4", seed=77)   # D4: a late spike key (rescale path)
    be = pkg.HipBackend(dev)
    sa = pkg.ShardedAttention(be, 0, 1, dist, force_collectives=True, merge=merge)
    assert sa.dist is dist
    sa.load_kv_shard_f64(torch.from_numpy(K).to(dev), torch.from_numpy(V).to(dev), n, dk, dv)
    Qd = torch.from_numpy(Q).to(dev)
    outs = sa.forward_batches([sa.convert_q(Qd[i:i + B].contiguous()) for i in range(0, m, B)])
    got = torch.cat([be.cvt_f2d(c, dv) for c in outs]).cpu().numpy()
    np.save(out, got)
finally:
    dist.destroy_process_group()
'''


@pytest.mark.parametrize("merge", ["gather", "allreduce"])
def test_sharded_attention_over_a_one_rank_rccl_group(merge, tmp_path, O):
    out = str(tmp_path / "res.npy")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, PKG, merge, str(tmp_path / "store"), out],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    Q, K, V = O.make_inputs(300, 1000, 128, 128, "D4", seed=77)
    want = O.numpy_attention_f64(Q, K, V)
    tol = 5e-5 * max(1.0, float(np.abs(V).max()))
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= tol
