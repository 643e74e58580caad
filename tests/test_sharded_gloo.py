"""CPU, world_size > 1 over gloo: the K/V-shard choreography of engine.attention_mpi /
ShardedAttention (dims broadcast, shard scatter, Q batch broadcast, all-reduce MAX,
all-reduce SUM, async reduce, root writeback -- attention-mpi.c:191-407) with a
checker-backed stand-in for the HIP stages (tests/_oracle_backend.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT, fp32_tol


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, dist_name, q_batch, out_path, mode="kv"):
    import importlib
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle as O
    from _oracle_backend import OracleBackend
    pkg = importlib.import_module(PKG)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n, dk, dv = shape
        if mode == "qrows":
            if rank == 0:
                Q, K, V = O.make_inputs(m, n, dk, dv, dist_name, seed=21)
                np.save(out_path, pkg.attention_qrows(Q, K, V, m, n, dk, dv, rank, world, dist=dist,
                                                      backend=OracleBackend()))
            else:
                assert pkg.attention_qrows(None, None, None, -1, -1, -1, -1, rank, world, dist=dist,
                                           backend=OracleBackend()) is None
        elif rank == 0:
            Q, K, V = O.make_inputs(m, n, dk, dv, dist_name, seed=21)
            res = pkg.attention_mpi(Q, K, V, m, n, dk, dv, rank, world, dist=dist,
                                    backend=OracleBackend(), q_batch=q_batch,
                                    merge="gather" if mode == "kv-gather" else "allreduce")
            np.save(out_path, res)
        else:   # non-root ranks hold nothing and pass garbage dims (attention-mpi.c:508-517)
            res = pkg.attention_mpi(None, None, None, -1, -1, -1, -1, rank, world, dist=dist,
                                    backend=OracleBackend(), q_batch=q_batch,
                                    merge="gather" if mode == "kv-gather" else "allreduce")
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,dist_name,q_batch", [
    (2, (70, 130, 72, 40), "D2", 32),     # 3 Q batches, ragged last one, odd shard split
    (2, (40, 257, 64, 64), "D4", 1000),   # one batch; late spike key lands in rank 1's shard
    (3, (16, 2, 8, 8), "D2", 8),          # n < world: rank 2 owns an empty shard
])
def test_attention_mpi_over_gloo(world, shape, dist_name, q_batch, tmp_path, orc, O):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), shape, dist_name, q_batch, out), nprocs=world, join=True)
    got = np.load(out)
    m, n, dk, dv = shape
    Q, K, V = O.make_inputs(m, n, dk, dv, dist_name, seed=21)
    want = orc.attention_f64(Q, K, V)
    assert got.shape == want.shape and np.isfinite(got).all()
    assert np.abs(got - want).max() <= fp32_tol(V)
    # and it reproduces the single-process restatement of the same pipeline to fp32 rounding
    same = orc.attention_sharded_f32(Q, K, V, world)
    assert np.abs(got - same).max() <= 1e-5 * max(1.0, np.abs(V).max())


@pytest.mark.parametrize("world,shape", [(2, (71, 130, 72, 40)), (3, (2, 50, 16, 8))])
def test_attention_qrows_over_gloo(world, shape, tmp_path, orc, O):
    """the Q-row-sharded alternative plan (no merge collective); world 3 with m=2 leaves a rank idle"""
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), shape, "D2", 0, out, "qrows"), nprocs=world, join=True)
    got = np.load(out)
    m, n, dk, dv = shape
    Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=21)
    want = orc.attention_f64(Q, K, V)
    assert got.shape == want.shape and np.abs(got - want).max() <= fp32_tol(V)


@pytest.mark.parametrize("world,shape,q_batch", [(2, (70, 130, 72, 40), 32), (3, (16, 2, 8, 8), 8)])
def test_attention_mpi_gather_merge_over_gloo(world, shape, q_batch, tmp_path, orc, O):
    """the one-all-gather variant of the shard merge (same algebra as attention-mpi.c:340-362)"""
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), shape, "D2", q_batch, out, "kv-gather"), nprocs=world, join=True)
    got = np.load(out)
    m, n, dk, dv = shape
    Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=21)
    assert np.abs(got - orc.attention_f64(Q, K, V)).max() <= fp32_tol(V)
