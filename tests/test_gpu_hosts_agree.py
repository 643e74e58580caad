"""GPU (-m gpu): the two hosts of the K/V-sharded path are ONE schedule.

  * the Python host -- one process per rank, `ShardedAttention(merge="gather", egress="scatter")`:
    fused kernel on the rank's shard, all-gather of the (lmax, lsum) pairs + merge_gathered, reduce-SCATTER
    of the normalised contributions, every rank widens its own rows (what `bench.py --gpus N` times);
  * the C host -- one process, `sdpa_attention_f64` on P ranks (here: loopback ranks, $SDPA_VIRTUAL_GPUS=P):
    the same stages from `tail_batch` (csrc/sdpa_host.hip).

Same kernels, same split plan (both launch on a stream that leaves the same number of workgroup slots free -- 16 CUs'
worth when the call has a next batch to hide a collective tail under, none for a one-batch call -- so both take the
same stream-K cuts), same merge algebra, sums in rank order on both sides (the C host's loopback
collectives; the dev-mode gloo adapter of bench.py): the results must agree BIT FOR BIT, P in {2, 3, 8}, with
ragged shards and several Q batches -- and both within the fp32 tolerance of the fp64 oracle.  (The C host
streamed its K/V shard as one launch per chunk in rounds 1-4, a different -- equally exact -- summation order: the first
three cases pin it to one launch per batch with the chunk knobs.  Since round 5 the default first batch is ONE streamed
launch with the device-level launch's own splits: the last two cases run the C host on its defaults.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT, fp32_tol, knob_env

pytestmark = pytest.mark.gpu

RANK = r'''
import ctypes, importlib, os, sys
import numpy as np, torch, torch.distributed as dist
ROOT, PKG, store, out_dir = sys.argv[1:5]
rank, world, m, n, d, B, seed, reserve = (int(x) for x in sys.argv[5:13])
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle as O
import bench
pkg = importlib.import_module(PKG)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=world)
d_ = bench.HostStagedDist(dist)                 # collectives staged through host memory, summed in rank order
try:
    Q, K, V = O.make_inputs(m, n, d, d, "D2", seed=seed)
    be = pkg.HipBackend(dev)
    sp = ctypes.c_void_p()                      # the C host's compute stream for P > 1: `reserve` CUs' worth of slots stay free
    pkg._lib.check(pkg.load().sdpa_dev_stream_create(reserve, ctypes.byref(sp)), "sdpa_dev_stream_create")
    with torch.cuda.stream(torch.cuda.ExternalStream(sp.value, device=dev)):
        sa = pkg.ShardedAttention(be, rank, world, d_, merge="gather", egress="scatter")
        c0, cn = pkg.owner_disp(n, world, rank), pkg.owner_count(n, world, rank)
        sa.load_kv_shard_f64(torch.from_numpy(K[c0:c0 + cn].copy()).to(dev), torch.from_numpy(V[c0:c0 + cn].copy()).to(dev), n, d, d)
        Qd = torch.from_numpy(Q).to(dev)
        for b, i0 in enumerate(range(0, m, B)):
            qf = sa.convert_q(Qd[i0:i0 + B].contiguous())
            contrib, lmax, lsum = sa.batch_partial(qf)
            rows, work, nrows = sa.batch_merge_egress(contrib, lmax, lsum)
            got = be.cvt_f2d(rows[:nrows], d).cpu().numpy() if nrows > 0 else np.zeros((0, d))
            np.save(os.path.join(out_dir, "b%d_r%d.npy" % (b, rank)), got)
    d_.barrier()
finally:
    dist.destroy_process_group()
'''

C_HOST = r'''
import importlib, os, sys
import numpy as np
ROOT, PKG, out = sys.argv[1:4]
m, n, d, seed, cus, streamed = (int(x) for x in sys.argv[4:10])
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle as O
pkg = importlib.import_module(PKG)
Q, K, V = O.make_inputs(m, n, d, d, "D2", seed=seed)
res = pkg.attention(Q, K, V)
t = pkg.last_timing()
assert t["virtual_ranks"] == 1 and t["egress"] == 2 and t["merge"] == 1 and t["compute_cus"] == cus, t
assert t["streamed"] == streamed, t
np.save(out, res)
'''


@pytest.mark.parametrize("world,m,n,B,streamed", [
    (2, 4096, 8192, 4096, 0),       # one batch, even shards
    (3, 5000, 10000, 2048, 0),      # ragged shards (3334, 3333, 3333), 3 batches, ragged shares
    (8, 4096, 16385, 4096, 0),      # 8 ranks, shards of 2049 / 2048 rows
    # round 5: the C host on its DEFAULT knobs -- every rank's first batch is one streamed launch fed by host converts --
    # is the Python host's device-level launch on the resident shard, bit for bit
    (2, 8192, 32768, 8192, 1),
    (3, 8192, 50000, 8192, 1)])     # ragged shards 16667 / 16667 / 16666: ragged last split and tile on every rank
def test_python_host_and_c_host_agree_bit_for_bit(world, m, n, B, streamed, tmp_path, O):
    d, seed = 128, 40 + world
    out_dir = str(tmp_path)
    store = str(tmp_path / "store")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, "-c", RANK, ROOT, PKG, store, out_dir] +
                              [str(x) for x in (r, world, m, n, d, B, seed, 16 if m > B else 0)], stderr=subprocess.PIPE,
                             text=True, env=env)
             for r in range(world)]
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2500:]
    py_rows = []
    for b, i0 in enumerate(range(0, m, B)):
        bs = min(B, m - i0)
        share = (bs + world - 1) // world
        parts = [np.load(os.path.join(out_dir, "b%d_r%d.npy" % (b, r))) for r in range(world)]
        assert [p.shape[0] for p in parts] == [max(0, min(share, bs - r * share)) for r in range(world)]
        py_rows.append(np.concatenate(parts))
    py = np.concatenate(py_rows)

    c_out = str(tmp_path / "c_host.npy")
    if streamed:
        # (the persistent launch with its K/V groups as ROW RANGES in key order, round 5's form: the interleaved groups of round 6 hold
        #  the shard's keys in another order -- the same sums in another order, equal within the tolerance, not bit for bit)
        cenv = dict(env, SDPA_VIRTUAL_GPUS=str(world), SDPA_QBATCH=str(B), SDPA_DEBUG="stream_interleave=0")
    else:
        cenv = dict(env, **knob_env(dict(SDPA_VIRTUAL_GPUS=str(world), SDPA_QBATCH=str(B), SDPA_ROW_PIECES="1",
                                         SDPA_KV_CHUNK_MIN=str(1 << 22), SDPA_KV_CHUNK_MAX=str(1 << 22), SDPA_HOST_CVT="0")))
    # 16 compute units' worth of workgroup slots stay free for the comm streams when the call has a NEXT batch to hide a
    # collective tail under (csrc/sdpa_host.hip: comm_cus_reserved, make_plan); a one-batch call gets the whole chip
    cus = 240 if m > B else 256
    r = subprocess.run([sys.executable, "-c", C_HOST, ROOT, PKG, c_out] + [str(x) for x in (m, n, d, seed, cus, streamed)],
                       capture_output=True, text=True, timeout=600, env=cenv)
    assert r.returncode == 0, r.stderr[-2500:]
    c = np.load(c_out)

    Q, K, V = O.make_inputs(m, n, d, d, "D2", seed=seed)
    want = O.numpy_attention_f64(Q, K, V)
    for name, got in (("python host", py), ("C host", c)):
        assert got.shape == want.shape and np.isfinite(got).all(), name
        assert np.abs(got - want).max() <= fp32_tol(V), name
    assert np.array_equal(py, c), "the two hosts differ in %d of %d values (max %.3e)" % (
        (py != c).sum(), py.size, np.abs(py - c).max())
