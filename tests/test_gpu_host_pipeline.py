"""GPU (-m gpu): the host-level pipeline of sdpa_attention_f64 (csrc/sdpa_host.hip) -- K/V chunk
streaming, the pieces of the last batch, and the P > 1 choreography of attention-mpi.c:340-399 run
on ONE device through loopback ranks (SDPA_VIRTUAL_GPUS=P), plus the real RCCL calls on a one-rank
communicator ($SDPA_DEBUG force_collectives=1).

Tolerance: max|got - fp64 oracle| <= 5e-5 * max(1, max|V|) for fp32
compute, 1e-2 * max(1, max|V|) for the bf16 path; NaN/Inf anywhere fails."""
import os

import numpy as np
import pytest
import torch

from conftest import fp32_tol, knob_env, set_knobs

pytestmark = pytest.mark.gpu


def check(got, want, V, what="", tol=None):
    tol = fp32_tol(V) if tol is None else tol
    assert got.shape == want.shape
    assert np.isfinite(got).all(), what + ": non-finite values"
    err = np.abs(got - want).max()
    assert err <= tol, "%s: max|err| %.3e > %.3e" % (what, err, tol)
    return err


def n_pieces(rows, row_pieces=4, min_rows=128):
    """row pieces of a batch as csrc/sdpa_host.hip plans them (whole query blocks of 128 rows)"""
    if row_pieces <= 1:
        return 1
    pr = max(-(-rows // row_pieces), min_rows)
    pr = -(-pr // 128) * 128
    return 1 if pr >= rows else -(-rows // pr)


@pytest.fixture
def engine(pkg, monkeypatch):
    """re-creates the engine with the environment a test asks for, and puts the default
    one-GPU engine back afterwards"""
    assert torch.cuda.is_available(), "the -m gpu tests need a real MI355X"

    def make(**env):
        pkg.shutdown()
        for k in ("SDPA_VIRTUAL_GPUS", "SDPA_QBATCH", "SDPA_PLAN", "SDPA_MERGE", "SDPA_PRECISION", "SDPA_EGRESS",
                  "SDPA_HOST_CVT", "SDPA_HOST_CVT_THREADS", "SDPA_COMM_CUS", "SDPA_HOST_WIDEN", "SDPA_HOST_REGISTER", "SDPA_STREAMED",
                  "SDPA_STREAM_TIMEOUT_MS", "SDPA_DEBUG"):
            monkeypatch.delenv(k, raising=False)
        for k, v in knob_env(env).items():          # (round-1..5 variable names of test knobs go into ONE $SDPA_DEBUG string)
            monkeypatch.setenv(k, v)
        pkg.init(1)
        return pkg

    yield make
    pkg.shutdown()
    for k in ("SDPA_VIRTUAL_GPUS", "SDPA_DEBUG"):
        monkeypatch.delenv(k, raising=False)
    pkg.init(1)


# ---------------------------------------------------------------- K/V chunk streaming ---------
@pytest.mark.parametrize("m,n,dk,dv,dist,prec", [
    (700, 9000, 128, 128, "D2", None),      # pipelined kernel, ragged rows and last chunk
    (513, 7000, 64, 64, "D4", None),        # late spike key sits in the last chunk
    (300, 5000, 72, 40, "D3", None),        # register-staged kernel (padded dims), peaky
    (200, 6000, 300, 96, "D2", None),       # dk-split kernel
    (140, 5000, 100, 200, "D2", None),      # fp32 dv > 128: two dv chunks per launch, every chunk into the slots
    (20, 3000, 600, 48, "D2", None),        # dk > 512: the VALU any-shape kernel (one split per launch)
    (300, 5000, 512, 200, "D1", "bf16"),    # bf16 general kernel (dk = 512, dv <= 256): no redo flags
    (700, 9000, 128, 128, "D2", "bf16"),    # bf16 pipe kernel: chunks of the transposed Vt image
    (260, 5000, 512, 512, "D1", "bf16"),    # bf16 wide kernel (redo flags live beside the slots)
])
def test_kv_chunk_streaming_matches_oracle(m, n, dk, dv, dist, prec, engine, orc, O):
    """the first Q batch starts on K/V chunk 0 while the later chunks are still crossing PCIe; the
    partial triples of all chunks and in-launch splits are merged in one pass; the last chunk runs
    in pieces whose finish + D2H overlap the next piece's kernel"""
    pkg = engine(SDPA_KV_CHUNK_MIN=1024, SDPA_KV_CHUNK_MAX=2048, SDPA_PIECE_MIN_ROWS=128)
    Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=m + n)
    want = orc.attention_f64(Q, K, V)
    tol = 1e-2 * max(1.0, float(np.abs(V).max())) if prec == "bf16" else None
    got = pkg.attention(Q, K, V, precision=prec)
    t = pkg.last_timing()
    assert t["kv_chunks"] >= 3 and t["q_batches"] == 1, t
    assert t["fused_launches"] == t["kv_chunks"] - 2 + 2 * n_pieces(m), t      # first and last chunk in row pieces
    # what ran is what the GPU-free planner (tests/test_host_plan.py) says would run
    pl = pkg.plan(m, n, dk, dv, 2 if prec == "bf16" else 0, 1)
    assert t["kv_chunks"] == len(pl["r"][0]["chunks"]) and t["q_batches"] == pl["q_batches"]
    check(got, want, V, "streamed", tol)
    # same problem, nothing streamed and nothing cut in pieces: must agree to rounding
    got1 = pkg.attention(Q, K, V, flags=1, precision=prec)      # SDPA_F_NO_PIPELINE
    t1 = pkg.last_timing()
    assert t1["kv_chunks"] == 1 and t1["fused_launches"] == 1, t1
    check(got1, want, V, "unstreamed", tol)
    assert np.abs(got - got1).max() <= (tol if tol else 2 * fp32_tol(V))
    # several Q batches on top: batch 0 streams, the later ones run on the resident shard
    pkg = engine(SDPA_KV_CHUNK_MIN=1024, SDPA_KV_CHUNK_MAX=2048, SDPA_QBATCH=256, SDPA_PIECE_MIN_ROWS=128)
    got2 = pkg.attention(Q, K, V, precision=prec)
    t2 = pkg.last_timing()
    assert t2["q_batches"] == (m + 255) // 256 and t2["kv_chunks"] >= 3
    check(got2, want, V, "streamed + batches", tol)


def test_streaming_is_deterministic(engine, O):
    pkg = engine(SDPA_KV_CHUNK_MIN=1024, SDPA_KV_CHUNK_MAX=4096)
    Q, K, V = O.make_inputs(1024, 12000, 128, 128, "D2", seed=4)
    a = pkg.attention(Q, K, V)
    for _ in range(5):
        assert np.array_equal(pkg.attention(Q, K, V), a), "same inputs must give bit-identical results"


def test_row_pieces_cover_ragged_rows(engine, orc, O):
    """row pieces (Q arriving / rows leaving) are whole query blocks; the last piece is ragged;
    with one K/V chunk the same launches serve as head and tail pieces"""
    for m in (129, 1000, 1025):
        for pieces in (1, 4, 8):
            for n in (3000, 900):
                pkg = engine(SDPA_ROW_PIECES=pieces, SDPA_PIECE_MIN_ROWS=128, SDPA_KV_CHUNK_MIN=1024,
                             SDPA_KV_CHUNK_MAX=1024)
                Q, K, V = O.make_inputs(m, n, 64, 64, "D2", seed=m)
                check(pkg.attention(Q, K, V), orc.attention_f64(Q, K, V), V, "m=%d n=%d pieces=%d" % (m, n, pieces))
                t = pkg.last_timing()
                want_pieces = n_pieces(m, pieces)
                assert t["fused_launches"] == (t["kv_chunks"] - 2 + 2 * want_pieces if t["kv_chunks"] > 1 else want_pieces), t


# ---------------------------------------------------------------- P > 1 on one device ---------
CASES_P = [
    # m,   n,    dk,  dv, dist, qbatch
    (96,  1000,  64,  64, "D4", 0),       # late spike key: the running max jumps in the LAST shard
    (64,     5,  16,  16, "D2", 0),       # n < P: empty shards (attention-mpi.c:172-173)
    (1000, 600,  64,  64, "D2", 192),     # 6 batches, ragged last one
    (300, 4100, 128, 128, "D3", 128),     # peaky, pipelined kernel, 3 batches
    (130,  700, 300,  72, "D2", 0),       # dk-split kernel
]


@pytest.mark.parametrize("P", [2, 3, 8])
@pytest.mark.parametrize("merge", ["gather", "allreduce"])
def test_virtual_ranks_kv_sharded(P, merge, engine, orc, O):
    """the C host's P > 1 branch (attention-mpi.c:340-399): owner_count/owner_disp shards, per-rank
    fused launches, all-gather or all-reduce(MAX)+all-reduce(SUM), reduce(SUM) to rank 0 -- vs the
    fp64 oracle and vs the restated fp32 pipeline of the reference at the same P"""
    for (m, n, dk, dv, dist, qb) in CASES_P:
        env = dict(SDPA_VIRTUAL_GPUS=P, SDPA_KV_CHUNK_MIN=1024, SDPA_KV_CHUNK_MAX=1024, SDPA_PIECE_MIN_ROWS=128)
        if qb:
            env["SDPA_QBATCH"] = qb
        pkg = engine(**env)
        Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=P + m)
        want = orc.attention_f64(Q, K, V)
        got = pkg.attention(Q, K, V, merge=merge)
        t = pkg.last_timing()
        assert t["n_gpus"] == P and t["virtual_ranks"] == 1 and t["plan"] == 0
        assert t["merge"] == (1 if merge == "gather" else 2)
        if qb:
            assert t["q_batches"] == (m + qb - 1) // qb
        check(got, want, V, "P=%d %s %s" % (P, merge, (m, n, dk, dv, dist)))
        ref32 = orc.attention_sharded_f32(Q, K, V, P)
        assert np.abs(got - ref32).max() <= 2 * fp32_tol(V), "vs the reference's fp32 pipeline at P=%d" % P


@pytest.mark.parametrize("P", [2, 3, 8])
def test_virtual_ranks_qrow_sharded(P, engine, orc, O):
    """SDPA_PLAN=qrows in the C host (attention-mpi.c:307-338 rows are independent): every rank
    holds all of K/V and finishes its own slice of the query rows; no collective"""
    for (m, n, dk, dv, dist, qb) in [(100, 900, 64, 64, "D2", 0), (5, 300, 32, 32, "D1", 0),
                                      (1000, 2500, 128, 128, "D3", 96)]:
        env = dict(SDPA_VIRTUAL_GPUS=P, SDPA_KV_CHUNK_MIN=1024, SDPA_KV_CHUNK_MAX=1024, SDPA_PIECE_MIN_ROWS=128)
        if qb:
            env["SDPA_QBATCH"] = qb
        pkg = engine(**env)
        Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=3 * P + m)
        got = pkg.attention(Q, K, V, plan="qrows")
        t = pkg.last_timing()
        assert t["plan"] == 1 and t["merge"] == 0 and t["n_gpus"] == P
        check(got, orc.attention_f64(Q, K, V), V, "qrows P=%d m=%d" % (P, m))


def test_virtual_ranks_bf16(engine, orc, O):
    pkg = engine(SDPA_VIRTUAL_GPUS=3, SDPA_KV_CHUNK_MIN=1024, SDPA_KV_CHUNK_MAX=1024, SDPA_QBATCH=256, SDPA_PIECE_MIN_ROWS=128)
    for (m, n, d) in [(600, 5000, 128), (130, 4, 64), (260, 3100, 512)]:
        Q, K, V = O.make_inputs(m, n, d, d, "D2", seed=m)
        got = pkg.attention(Q, K, V, precision="bf16")
        check(got, orc.attention_f64(Q, K, V), V, "bf16 P=3 %s" % ((m, n, d),), 1e-2 * max(1.0, float(np.abs(V).max())))


def test_rccl_calls_on_a_one_rank_communicator(engine, orc, O):
    """SDPA_FORCE_COLLECTIVES=1: the merge branch with the REAL RCCL entry points (dlopen'd
    ncclCommInitAll / ncclAllGather / ncclAllReduce / ncclReduce, grouped) on a communicator of
    one rank -- argument order, datatypes and stream ordering of sdpa_coll.hip's RCCL side"""
    pkg = engine(SDPA_FORCE_COLLECTIVES=1, SDPA_QBATCH=300)
    Q, K, V = O.make_inputs(1000, 3000, 128, 128, "D2", seed=21)
    want = orc.attention_f64(Q, K, V)
    for merge, code in (("gather", 1), ("allreduce", 2)):
        got = pkg.attention(Q, K, V, merge=merge)
        t = pkg.last_timing()
        assert t["merge"] == code and t["n_gpus"] == 1 and t["virtual_ranks"] == 0 and t["q_batches"] == 4
        check(got, want, V, "rccl one rank, " + merge)


def test_kv_prefetch_then_compute(engine, orc, O):
    """sdpa_kv_prefetch (reading K/V and moving them overlap): rows announced in
    pieces, K first then V as the file stores them; the following compute call skips what is staged
    and must give the same bits as a call without prefetch; a prefetch of other arrays is void"""
    pkg = engine(SDPA_KV_CHUNK_MIN=1024, SDPA_KV_CHUNK_MAX=2048, SDPA_PIECE_MIN_ROWS=128)
    lib = pkg.load()
    m, n, dk, dv = 700, 9000, 128, 128
    Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=61)
    want = orc.attention_f64(Q, K, V)
    plain = pkg.attention(Q, K, V)
    check(plain, want, V, "no prefetch")
    Kc, Vc = np.ascontiguousarray(K), np.ascontiguousarray(V)
    for r in range(0, n, 1500):
        assert lib.sdpa_kv_prefetch(Kc.ctypes.data, Vc.ctypes.data, m, n, dk, dv, 0, min(n, r + 1500), 0) == 0
    for r in range(0, n, 2100):
        assert lib.sdpa_kv_prefetch(Kc.ctypes.data, Vc.ctypes.data, m, n, dk, dv, 0, n, min(n, r + 2100)) == 0
    got = pkg.attention(Q, Kc, Vc)
    assert np.array_equal(got, plain), "prefetched K/V must give the same bits"
    # partial prefetch (K only, half of it), then compute
    assert lib.sdpa_kv_prefetch(Kc.ctypes.data, Vc.ctypes.data, m, n, dk, dv, 0, n // 2, 0) == 0
    assert np.array_equal(pkg.attention(Q, Kc, Vc), plain)
    # a prefetch for different arrays / dims must not leak into this call
    K2, V2 = K * 0.5, V + 1.0
    assert lib.sdpa_kv_prefetch(K2.ctypes.data, V2.ctypes.data, m, n, dk, dv, 0, n, n) == 0
    assert np.array_equal(pkg.attention(Q, Kc, Vc), plain)
    assert lib.sdpa_kv_prefetch(Kc.ctypes.data, Vc.ctypes.data, m, n, dk, dv, 0, n + 1, 0) == pkg._lib.SDPA_EINVAL
    # bf16 image and virtual ranks
    pkg = engine(SDPA_VIRTUAL_GPUS=3, SDPA_KV_CHUNK_MIN=1024, SDPA_KV_CHUNK_MAX=1024, SDPA_PIECE_MIN_ROWS=128)
    lib = pkg.load()
    plain = pkg.attention(Q, Kc, Vc, precision="bf16")
    assert lib.sdpa_kv_prefetch(Kc.ctypes.data, Vc.ctypes.data, m, n, dk, dv, 2, n, n - 1000) == 0
    assert np.array_equal(pkg.attention(Q, Kc, Vc, precision="bf16"), plain)


def test_engine_restores_the_callers_device_and_survives_reinit(engine, O, orc):
    pkg = engine()
    torch.cuda.set_device(0)
    Q, K, V = O.make_inputs(64, 256, 32, 32, "D1", seed=1)
    want = orc.attention_f64(Q, K, V)
    for env in ({"SDPA_VIRTUAL_GPUS": 2}, {}, {"SDPA_VIRTUAL_GPUS": 5}, {}):
        pkg = engine(**env)
        check(pkg.attention(Q, K, V), want, V, str(env))
        assert torch.cuda.current_device() == 0
        assert pkg.last_timing()["n_gpus"] == int(env.get("SDPA_VIRTUAL_GPUS", 1))


def test_tune_env_cannot_change_the_shipped_library(engine, O, orc, monkeypatch):
    """$SDPA_TUNE selected timing-only ablation kernels in round 1; the shipped build must ignore it"""
    set_knobs(monkeypatch, SDPA_TUNE=112)
    pkg = engine()
    Q, K, V = O.make_inputs(300, 2000, 128, 128, "D2", seed=2)
    check(pkg.attention(Q, K, V), orc.attention_f64(Q, K, V), V, "SDPA_TUNE=112")
    Q, K, V = O.make_inputs(130, 1000, 512, 512, "D1", seed=2)
    set_knobs(monkeypatch, SDPA_TUNE=15 << 8)
    check(pkg.attention(Q, K, V, precision="bf16"), orc.attention_f64(Q, K, V), V, "SDPA_TUNE bf16", 1e-2 * max(1.0, float(np.abs(V).max())))


# ------------------------------------------------- P > 1: enqueue threads, comm streams, egress ---------
@pytest.mark.parametrize("P,m,n,d,batch", [(2, 1500, 9000, 128, 512), (3, 700, 6000, 64, 256), (8, 2048, 40000, 128, 1024),
                                           (8, 300, 5, 64, 128)])           # n < P: three ranks own no key at all
def test_multi_rank_schedules_agree_bit_for_bit(P, m, n, d, batch, engine, orc, O):
    """P loopback ranks, several Q batches.  The schedule of round 3 -- one enqueue thread per rank, the
    per-batch collectives on the ranks' comm streams (batch b's merge under batch b+1's kernels,
    attention-mpi.c:364-380), the merged rows reduce-SCATTERED so that every rank sends its share home --
    against the schedule of round 2 (one thread, reduce to the root, $SDPA_EGRESS=root
    $SDPA_DEBUG=enqueue_threads=0): the loopback collectives sum in rank order either way, so the results must
    be IDENTICAL bit for bit, both merges; and within tolerance of the fp64 oracle."""
    Q, K, V = O.make_inputs(m, n, d, d, "D4", seed=P * 1000 + m)
    want = orc.attention_f64(Q, K, V)
    common = dict(SDPA_VIRTUAL_GPUS=P, SDPA_QBATCH=batch, SDPA_KV_CHUNK_MIN=1024, SDPA_KV_CHUNK_MAX=2048)
    for merge in ("gather", "allreduce"):
        pkg = engine(SDPA_EGRESS="root", SDPA_ENQUEUE_THREADS=0, SDPA_MERGE=merge, **common)
        old = pkg.attention(Q, K, V)
        t_old = pkg.last_timing()
        assert t_old["enqueue_threads"] == 1 and t_old["egress"] == 1 and t_old["n_gpus"] == P, t_old
        check(old, want, V, "round-2 schedule, %s" % merge)
        # (the registered paths of rounds 1-3 -- opt-in since round 4 -- are covered in a process of their own,
        #  tests/test_gpu_register_optin.py: a registration poisons the process for PyTorch's pageable copies)
        for knobs in (dict(), dict(SDPA_EGRESS="root"), dict(SDPA_ENQUEUE_THREADS=0), dict(SDPA_HOST_CVT=0, SDPA_HOST_WIDEN=0)):
            pkg = engine(SDPA_MERGE=merge, **common, **knobs)
            for rep in range(3):
                new = pkg.attention(Q, K, V)
                assert np.array_equal(new, old), "merge %s, knobs %r, call %d: differs from the one-thread / root-egress result" % (merge, knobs, rep)
            t = pkg.last_timing()
            assert t["enqueue_threads"] == (1 if knobs.get("SDPA_ENQUEUE_THREADS") == 0 else P), t
            assert t["egress"] == (1 if knobs.get("SDPA_EGRESS") == "root" else 2), t
            assert t["q_batches"] == -(-m // batch) and t["merge"] == (1 if merge == "gather" else 2), t
            first = t["enqueue_first_kernel_us"]
            assert len(first) == P and all(x > 0 for x in first), t


def test_enqueue_threads_issue_every_ranks_first_kernel_together(engine, O):
    """config 3's shape on 8 loopback ranks (n = 262144 sharded 8 ways, m reduced): with one enqueue thread per
    rank, rank 7's first fused launch is ENQUEUED within a fraction of a millisecond of rank 0's (host clock,
    hardware independent); with the single enqueue thread of round 2 it waited behind seven ranks' worth
    of API calls.  Logged for profiles/; the bound asserted here is deliberately loose.
    (Caller arrays from sdpa_host_alloc and device converts: the enqueue path alone.  With pageable arrays -- host
    converts since round 4 -- a rank's first launch also waits for the pool to have converted ITS first chunk, which
    comes behind the lower ranks' in the pool's queue: a property of the feed, not of who enqueues.)"""
    import ctypes
    m, n, d = 4096, 262144, 128
    rng = np.random.default_rng(5)
    lib = engine().load()
    bufs = []
    def pinned(shape):
        a = rng.uniform(-1, 1, shape)
        p = lib.sdpa_host_alloc(a.nbytes)
        assert p
        bufs.append(p)
        out = np.ctypeslib.as_array((ctypes.c_double * a.size).from_address(p)).reshape(a.shape)
        out[...] = a
        return out
    try:
        Q, K, V = pinned((m, d)), pinned((n, d)), pinned((n, d))
        spreads = {}
        for threads in (0, 1):
            pkg = engine(SDPA_VIRTUAL_GPUS=8, SDPA_ENQUEUE_THREADS=threads, SDPA_HOST_CVT=0)
            best = None
            for _ in range(4):
                pkg.attention(Q, K, V)
                f = pkg.last_timing()["enqueue_first_kernel_us"]
                sp = max(f) - min(f)
                best = sp if best is None else min(best, sp)
            spreads[threads] = best
            print("first-kernel enqueue spread over 8 ranks, enqueue threads %s: %.0f us" % ("on" if threads else "off", best))
        del Q, K, V
    finally:
        for p in bufs:
            lib.sdpa_host_free(p)
    assert spreads[1] < 1000.0, spreads
    assert spreads[1] < spreads[0], spreads


def test_one_rank_forced_collectives_use_the_comm_stream(engine, orc, O):
    """one rank with the collectives forced on (a one-rank RCCL communicator): the tail of each batch runs on
    the comm stream behind the rank's kernels; 5 batches"""
    pkg = engine(SDPA_FORCE_COLLECTIVES=1, SDPA_QBATCH=256)
    Q, K, V = O.make_inputs(1200, 3000, 128, 128, "D2", seed=12)
    for merge in ("gather", "allreduce"):
        got = pkg.attention(Q, K, V, merge=merge)
        t = pkg.last_timing()
        assert t["q_batches"] == 5 and t["merge"] == (1 if merge == "gather" else 2) and t["egress"] == 1, t
        check(got, orc.attention_f64(Q, K, V), V, "forced collectives, %s" % merge)


# ------------------------------------------------- $SDPA_HOST_CVT: the reference's own convert placement -----
@pytest.mark.parametrize("m,n,dk,dv,prec,env", [
    (700, 9000, 128, 128, None, {}),                                   # dense fp32 images, streamed chunks
    (300, 5000, 72, 40, None, {}),                                     # padded fp32 images (ld 128 / 64)
    (200, 6000, 300, 96, None, {"SDPA_QBATCH": 64}),                   # dk-split kernel, 4 Q batches
    (700, 9000, 128, 128, "bf16", {}),                                 # bf16: K/Q images from the host, V rows transposed on the device
    (260, 5000, 512, 512, "bf16", {}),                                 # bf16 wide kernel
    (300, 5000, 100, 200, "bf16", {"SDPA_QBATCH": 128}),               # bf16 padded dims, 3 batches
    (1500, 9000, 128, 128, None, {"SDPA_VIRTUAL_GPUS": 3, "SDPA_QBATCH": 512}),                      # 3 loopback ranks, 3 batches
    (900, 7000, 64, 64, None, {"SDPA_VIRTUAL_GPUS": 2, "SDPA_PLAN": "qrows"}),                       # query rows sharded: every rank converts its own Q rows
    (300, 5, 64, 64, None, {"SDPA_VIRTUAL_GPUS": 4}),                                                # n < P: empty shards
])
def test_host_side_convert_gives_the_device_converts_result_bit_for_bit(m, n, dk, dv, prec, env, engine, orc, O):
    """$SDPA_HOST_CVT=1: host threads convert fp64 -> fp32 / bf16 operand images into page-locked staging (the
    reference converts on the host too, cvt_d2f_avx512 at attention-mpi.c:224-225, :303) and half / a quarter
    of the bytes cross PCIe.  Same roundings as the device converters, so the SAME images and the same
    result bit for bit; checked against the fp64 oracle as well."""
    Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=m + dk)
    common = dict(SDPA_KV_CHUNK_MIN=1024, SDPA_KV_CHUNK_MAX=2048, SDPA_PIECE_MIN_ROWS=128, **env)
    pkg = engine(SDPA_HOST_CVT=0, **common)
    want = pkg.attention(Q, K, V, precision=prec)
    assert pkg.last_timing()["host_convert_threads"] == 0
    tol = 1e-2 * max(1.0, float(np.abs(V).max())) if prec == "bf16" else None
    check(want, orc.attention_f64(Q, K, V), V, "device converts", tol)
    for threads in (3, 16):
        pkg = engine(SDPA_HOST_CVT=1, SDPA_HOST_CVT_THREADS=threads, **common)
        for rep in range(2):
            got = pkg.attention(Q, K, V, precision=prec)
            assert np.array_equal(got, want), "host converts (%d threads, call %d) differ from the device converts" % (threads, rep)
        t = pkg.last_timing()
        assert t["host_convert_threads"] == threads and t["register_us"] >= 0, t


# ------------------------------------------------- $SDPA_HOST_WIDEN: the root widens the rows (attention-mpi.c:373, :396) -----
@pytest.mark.parametrize("m,n,dk,dv,prec,env", [
    (1000, 9000, 128, 128, None, {}),                                  # one rank: the last chunk's rows leave in pieces
    (300, 5000, 72, 40, None, {}),                                     # padded rows (ldo 64, dv 40): repacked on the device
    (700, 3000, 64, 64, None, {"SDPA_QBATCH": 256}),                   # 3 Q batches: the dense32 buffers rotate
    (260, 5000, 512, 512, "bf16", {}),                                 # bf16 wide kernel
    (1500, 9000, 128, 128, None, {"SDPA_VIRTUAL_GPUS": 3, "SDPA_QBATCH": 512}),                            # reduce-scatter egress
    (1500, 9000, 128, 128, None, {"SDPA_VIRTUAL_GPUS": 3, "SDPA_QBATCH": 512, "SDPA_EGRESS": "root"}),    # reduce to the root
    (700, 6000, 72, 40, None, {"SDPA_VIRTUAL_GPUS": 2, "SDPA_MERGE": "allreduce"}),                       # padded rows through the collectives
    (900, 7000, 64, 64, None, {"SDPA_VIRTUAL_GPUS": 2, "SDPA_PLAN": "qrows"}),                            # every rank finishes its own rows
    (300, 5, 64, 64, None, {"SDPA_VIRTUAL_GPUS": 4}),                                                      # n < P: empty shards
    (5, 700, 64, 64, None, {"SDPA_VIRTUAL_GPUS": 4}),                                                      # m < P: ranks with no row to send home
])
def test_host_side_widening_gives_the_device_result_bit_for_bit(m, n, dk, dv, prec, env, engine, orc, O):
    """$SDPA_HOST_WIDEN=1: the normalised rows cross PCIe as fp32 into page-locked staging and host threads widen them
    into `result` -- where the reference widens them (cvt_f2d_avx512 on the root, attention-mpi.c:373 / :396).  The
    device's fp32 value is the same and fp32 -> fp64 is exact: the same result bit for bit, in every egress of the
    pipeline, with `result` at any alignment (it is never registered in this mode)."""
    Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=m + dk)
    common = dict(SDPA_KV_CHUNK_MIN=1024, SDPA_KV_CHUNK_MAX=2048, SDPA_PIECE_MIN_ROWS=128, **env)
    pkg = engine(SDPA_HOST_WIDEN=0, **common)
    want = pkg.attention(Q, K, V, precision=prec)
    assert pkg.last_timing()["host_widen"] == 0
    tol = 1e-2 * max(1.0, float(np.abs(V).max())) if prec == "bf16" else None
    check(want, orc.attention_f64(Q, K, V), V, "device widening", tol)
    for threads in (1, 3, 16):
        pkg = engine(SDPA_HOST_WIDEN=1, SDPA_HOST_CVT_THREADS=threads, **common)
        for rep in range(2):
            got = pkg.attention(Q, K, V, precision=prec)
            assert np.array_equal(got, want), "host widening (%d threads, call %d) differs from the device's" % (threads, rep)
        t = pkg.last_timing()
        assert t["host_widen"] == 1 and t["tail_us"] >= 0, t
    # together with host-side input converts (one pool serves both)
    pkg = engine(SDPA_HOST_WIDEN=1, SDPA_HOST_CVT=1, SDPA_HOST_CVT_THREADS=8, **common)
    assert np.array_equal(pkg.attention(Q, K, V, precision=prec), want)


def test_widening_placement_is_chosen_per_problem(engine, O):
    """$SDPA_HOST_WIDEN unset: the host widens when it has the threads and the result is worth waking them for"""
    import os
    pkg = engine()
    rng = np.random.default_rng(2)
    Q, K, V = (rng.uniform(-1, 1, s) for s in ((4096, 128), (2048, 128), (2048, 128)))
    a = pkg.attention(Q, K, V)
    big_host = (os.cpu_count() or 1) >= 16
    assert pkg.last_timing()["host_widen"] == (1 if big_host else 0)
    assert pkg.attention(Q[:64], K, V).shape == (64, 128)                  # tiny result: the device widens
    assert pkg.last_timing()["host_widen"] == 0
    pkg = engine(SDPA_HOST_WIDEN=0)
    assert np.array_equal(pkg.attention(Q, K, V), a)


def test_convert_placement_is_chosen_per_problem(engine, orc, O):
    """$SDPA_HOST_CVT unset: per problem.  PAGEABLE caller arrays (numpy's) are never registered since round 4, so on a
    host with the threads for it they always go through the library's page-locked staging (host converts).  Arrays
    that ARE page-locked (sdpa_host_alloc, the CLI's reader) -- or registered on request, $SDPA_HOST_REGISTER=1 --
    take the round-3 model: host threads convert when the fp64 inputs would take clearly longer over PCIe than the
    kernels take (one rank only), the device converts when the kernels cover the transfer anyway."""
    import ctypes
    import os
    big_host = (os.cpu_count() or 1) >= 16
    pkg = engine()
    lib = pkg.load()
    Q, K, V = O.make_inputs(260, 5000, 512, 512, "D1", seed=3)            # config 5's dims: copy bound
    got = pkg.attention(Q, K, V, precision="bf16")
    assert (pkg.last_timing()["host_convert_threads"] > 0) == big_host
    assert pkg.last_timing()["register_us"] == 0                          # nothing is registered by default
    check(got, orc.attention_f64(Q, K, V), V, "auto -> host converts", 1e-2 * max(1.0, float(np.abs(V).max())))
    rng = np.random.default_rng(1)
    Q, K, V = (rng.uniform(-1, 1, s) for s in ((16384, 128), (16384, 128), (16384, 128)))   # kernel bound
    want = pkg.attention(Q, K, V)
    assert (pkg.last_timing()["host_convert_threads"] > 0) == big_host    # pageable: staged all the same
    assert pkg.attention(Q[:64], K[:512], V[:512]).shape == (64, 128)     # tiny: latency bound either way
    assert pkg.last_timing()["host_convert_threads"] == 0
    # the same kernel-bound problem from page-locked caller arrays: round 5 -- its first batch can run as ONE streamed launch,
    # which only the host converts can feed (no convert kernel runs beside a persistent launch): host converts again;
    # a kernel-bound problem whose kernels have no streamed form (d = 256) keeps the device converts
    bufs = []
    def pinned_copy(a):
        p = lib.sdpa_host_alloc(a.nbytes)
        assert p
        bufs.append(p)
        out = np.ctypeslib.as_array((ctypes.c_double * a.size).from_address(p)).reshape(a.shape)
        out[...] = a
        return out
    try:
        Qp, Kp, Vp = pinned_copy(Q), pinned_copy(K), pinned_copy(V)
        got = pkg.attention(Qp, Kp, Vp)
        t = pkg.last_timing()
        assert (t["host_convert_threads"] > 0) == big_host and t["streamed"] == (1 if big_host else 0) and t["register_us"] == 0, t
        assert np.array_equal(got, want)
        del Qp, Kp, Vp
        Q2, K2, V2 = (pinned_copy(rng.uniform(-1, 1, s)) for s in ((16384, 256), (16384, 256), (16384, 256)))
        got = pkg.attention(Q2, K2, V2)
        t = pkg.last_timing()
        assert t["host_convert_threads"] == 0 and t["streamed"] == 0 and t["register_us"] == 0, t
        assert np.isfinite(got).all()
        del Q2, K2, V2
    finally:
        for q in bufs:
            lib.sdpa_host_free(q)
    # (registration on request, $SDPA_HOST_REGISTER=1: tests/test_gpu_register_optin.py, in a process of its own)
    Q, K, V = O.make_inputs(260, 5000, 512, 512, "D1", seed=3)
    pkg = engine(SDPA_VIRTUAL_GPUS=2)                                      # pageable, two ranks: one pool serves both
    got = pkg.attention(Q, K, V, precision="bf16")
    assert (pkg.last_timing()["host_convert_threads"] > 0) == big_host
    check(got, orc.attention_f64(Q, K, V), V, "two ranks, host converts", 1e-2 * max(1.0, float(np.abs(V).max())))


def test_arrays_the_caller_pinned_itself_are_used_in_place_once_declared(engine, O):
    """page-locked memory the library did not allocate (here: pinned torch tensors) is taken for pageable -- the library
    does not ask the runtime about pointers it does not know -- until the caller declares it (sdpa_host_declare_pinned): then a
    kernel-bound problem without a streamed form (d = 256) takes the direct path (fp64 over the link, device converts), as from
    sdpa_host_alloc memory; the results are the same bits either way; forgetting the range brings the staging back."""
    import os
    big_host = (os.cpu_count() or 1) >= 16
    pkg = engine()
    lib = pkg.load()
    rng = np.random.default_rng(4)
    shapes = ((16384, 256), (16384, 256), (16384, 256))             # kernel bound (a copy-bound problem takes the host converts anyway)
    T = [torch.from_numpy(rng.uniform(-1, 1, s)).pin_memory() for s in shapes]
    Q, K, V = (t.numpy() for t in T)
    want = pkg.attention(Q, K, V)
    assert (pkg.last_timing()["host_convert_threads"] > 0) == big_host          # undeclared: staged like a pageable array
    try:
        for t in T:
            assert lib.sdpa_host_declare_pinned(t.data_ptr(), t.numel() * 8) == 0
        got = pkg.attention(Q, K, V)
        t = pkg.last_timing()
        assert t["host_convert_threads"] == 0 and t["streamed"] == 0 and t["register_us"] == 0, t
        assert np.array_equal(got, want)
    finally:
        for t in T:
            assert lib.sdpa_host_forget_pinned(t.data_ptr()) == 0
    assert lib.sdpa_host_forget_pinned(T[0].data_ptr()) < 0                     # never declared (any more)
    assert np.array_equal(pkg.attention(Q, K, V), want)
    assert (pkg.last_timing()["host_convert_threads"] > 0) == big_host


# ------------------------------------------------- the streamed first batch (round 5): ONE persistent launch that follows its inputs -----
def stream_halves(pkg, m, n, dk, dv, precision="f32"):
    """how many launches the streamed first batch of this problem is (sdpa_plan_describe): 1, or 2 half-row launches (round 6)"""
    sp = pkg.plan(m, n, dk, dv, 2 if precision == "bf16" else 0, 1)["r"][0]["stream"]
    return (sp["halves"], sp["rows_per_launch"]) if sp["on"] else (1, 0)


def stream_key_order(pkg, m, n, dk, dv, precision="f32"):
    """image row j of the streamed launch holds key order[j] of the shard: the identity, or -- interleaved groups (round 6) -- group c
    = the c-th contiguous key range, its `splits` slices at tiles [end_tile[c-1], end_tile[c]) of the splits' ranges"""
    sp = pkg.plan(m, n, dk, dv, 2 if precision == "bf16" else 0, 1)["r"][0]["stream"]
    order = np.arange(n)
    if sp["on"] and sp["interleaved"]:
        tps, S = sp["tiles_per_split"], sp["splits"]
        for (k0, keys, c), a, b in zip(sp["entries"], [0] + sp["end_tile"], sp["end_tile"]):
            sl = keys // S
            assert sl == (b - a) * 32
            for sx in range(S):
                order[(sx * tps + a) * 32:(sx * tps + b) * 32] = np.arange(k0 + sx * sl, k0 + (sx + 1) * sl)
        assert np.array_equal(np.sort(order), np.arange(n))
    return order


def device_level(pkg, Q, K, V, batch, precision="f32"):
    """the device-level path on resident inputs, batch after batch: converts, ONE fused launch per batch on the whole
    shard (sdpa_dev_shard_partial_f32 / _bf16), finish -- what the streamed launch must reproduce bit for bit.  Where the
    host runs the first batch as TWO half-row launches (round 6: the first half's rows leave under the second half's MFMAs),
    so does this: the same device-level launch per half."""
    be = pkg.HipBackend("cuda:0")
    sa = pkg.ShardedAttention(be, precision=precision)
    n, dk = K.shape
    dv = V.shape[1]
    order = stream_key_order(pkg, Q.shape[0], n, dk, dv, precision)      # (the keys in the order the host's images hold them)
    sa.load_kv_shard_f64(torch.from_numpy(np.ascontiguousarray(K[order])).cuda(), torch.from_numpy(np.ascontiguousarray(V[order])).cuda(), n, dk, dv)
    halves, rows_launch = stream_halves(pkg, Q.shape[0], n, dk, dv, precision)
    out = []
    for i0 in range(0, Q.shape[0], batch):
        rows = min(batch, Q.shape[0] - i0)
        step = rows_launch if (i0 == 0 and halves == 2) else rows
        for j0 in range(i0, i0 + rows, step):
            qf = sa.convert_q(torch.from_numpy(Q[j0:j0 + step]).cuda())
            contrib, lmax, lsum = sa.batch_partial(qf)
            out.append(be.finish_f64(contrib, lsum, dv).cpu().numpy())
    return np.concatenate(out)


@pytest.mark.parametrize("m,n,dk,dv,dist,env", [
    (8192, 8192, 128, 128, "D2", {}),                       # BASELINE config 2: 8 splits, 3 interleaved groups (one pitched copy each)
    (8192, 8192, 128, 128, "D4", {"SDPA_DEBUG": "stream_interleave=0"}),    # ... and round 5's form of it: one group
    (32768, 65536, 128, 128, "D2", {}),                     # the metric shape: 2 splits, 6 interleaved groups
    (16384, 20000, 128, 128, "D2", {}),                     # 4 splits of 157 tiles: ragged last split and last tile
    (32768, 16384, 64, 64, "D1", {}),                       # 2 splits, 64-wide images
    (8192, 12000, 100, 72, "D3", {}),                       # padded dims (images 128 wide), peaky scores
    (16384, 9001, 128, 64, "D4", {}),                       # adversarial: the spike key sits late in the last group; ragged last tile
    (20000, 16384, 128, 128, "D2", {"SDPA_QBATCH": 8192}),  # 3 batches: the first streamed, the others on the resident shard
    (8192, 40000, 128, 128, "D2", {"SDPA_STREAM_CHUNK_MIN": 1024, "SDPA_KV_CHUNK_MAX": 4096}),   # many small groups
])
def test_streamed_first_batch_is_the_device_level_launch_bit_for_bit(m, n, dk, dv, dist, env, engine, orc, O):
    """Round 5: with host converts feeding it, the first Q batch is ONE persistent launch -- the classic
    grid over the whole shard -- whose workgroups wait in the kernel for the ready word of the K/V group (and the Q row
    piece) they are about to read; the copy engine raises the words behind the bytes.  Same pieces of work and the same
    operations in the same order as sdpa_dev_shard_partial_f32 on resident inputs: the SAME result bit for bit (and so
    deterministic from call to call); the launch-per-chunk schedule ($SDPA_STREAMED=0) sums in another order and agrees
    within the path's tolerance; both against the fp64 oracle."""
    Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=m + n)
    batch = int(env.get("SDPA_QBATCH", 32768))
    pkg = engine(**env)
    got = pkg.attention(Q, K, V)
    t = pkg.last_timing()
    assert t["streamed"] == 1 and t["host_convert_threads"] > 0, t
    assert "fused_pipelined" in t["last_kernel"], t["last_kernel"]
    if m <= batch:
        assert t["fused_launches"] == stream_halves(pkg, m, n, dk, dv)[0] and t["last_kernel"].startswith("sdpa::fused_pipelined_stream_kernel<"), t
    want = device_level(pkg, Q, K, V, batch)
    assert np.array_equal(got, want), "streamed launch differs from the device-level launch in %d values (max %.3e)" % (
        (got != want).sum(), np.abs(got - want).max())
    for rep in range(2):
        assert np.array_equal(pkg.attention(Q, K, V), got), "call %d differs" % rep
    rows = np.sort(np.random.default_rng(m).choice(m, 48, replace=False))
    ref = O.numpy_attention_f64(Q, K, V, rows)
    check(got[rows], ref, V, "streamed")
    pkg = engine(SDPA_STREAMED=0, **env)
    old = pkg.attention(Q, K, V)
    assert pkg.last_timing()["streamed"] == 0 and pkg.last_timing()["fused_launches"] > 1
    check(old[rows], ref, V, "launch per chunk")
    assert np.abs(old - got).max() <= 2 * fp32_tol(V)


@pytest.mark.parametrize("m,n", [(8192, 8192), (8192, 32768), (16384, 8192)])
def test_streamed_calls_back_to_back_on_different_inputs_never_see_the_previous_calls_bytes(m, n, engine, O):
    """The operand images of a streamed call land in the SAME device buffers as the previous call's while the launch is
    already resident -- after the runtime's invalidate at the launch's start.  What keeps a workgroup from multiplying the
    previous call's K/V (or Q) out of an L2 / L1 line is the system-scope acquire behind each ready word.  Shapes whose
    images fit the 4 MiB L2s (config 2; 16384 x 8192: 4 splits) and a two-group one, two different input sets alternating
    in one engine: every call equals its own set's first result bit for bit and the fp64 restatement on a row subset."""
    d = 128
    sets = [O.make_inputs(m, n, d, d, dist, seed=seed) for dist, seed in (("D1", 11), ("D2", 12))]
    pkg = engine()
    rows = np.arange(0, m, max(1, m // 40))
    first = []
    for Q, K, V in sets:
        got = pkg.attention(Q, K, V)
        assert pkg.last_timing()["streamed"] == 1, pkg.last_timing()
        check(got[rows], O.numpy_attention_f64(Q, K, V, rows), V, "first call of a set")
        first.append(got)
    assert np.abs(first[0] - first[1]).max() > 1e-3               # (the sets really differ)
    for rep in range(6):
        for (Q, K, V), want in zip(sets, first):
            got = pkg.attention(Q, K, V)
            assert np.array_equal(got, want), "round %d: %d values differ from the set's first result (max %.3e)" % (
                rep, (got != want).sum(), np.abs(got - want).max())


def test_streamed_launch_from_page_locked_caller_arrays_and_on_loopback_ranks(engine, orc, O):
    """the CLI's arrays (sdpa_host_alloc) take the streamed form too -- the host converts feed it, a device convert
    could not run beside the persistent launch; so do the ranks of a K/V-sharded call, each on its own shard"""
    import ctypes
    m, n, d = 8192, 16384, 128
    Q, K, V = O.make_inputs(m, n, d, d, "D2", seed=77)
    pkg = engine()
    lib = pkg.load()
    want = pkg.attention(Q, K, V)
    bufs = []

    def pinned_copy(a):
        p = lib.sdpa_host_alloc(a.nbytes)
        assert p
        bufs.append(p)
        out = np.ctypeslib.as_array((ctypes.c_double * a.size).from_address(p)).reshape(a.shape)
        out[...] = a
        return out
    try:
        Qp, Kp, Vp = pinned_copy(Q), pinned_copy(K), pinned_copy(V)
        got = pkg.attention(Qp, Kp, Vp)
        t = pkg.last_timing()
        assert t["streamed"] == 1 and t["host_convert_threads"] > 0, t
        assert np.array_equal(got, want)
        del Qp, Kp, Vp
    finally:
        for q in bufs:
            lib.sdpa_host_free(q)
    pkg = engine(SDPA_VIRTUAL_GPUS=2)
    got2 = pkg.attention(Q, K, V)
    t = pkg.last_timing()
    assert t["streamed"] == 1 and t["n_gpus"] == 2, t
    rows = np.arange(0, m, 171)
    check(got2[rows], O.numpy_attention_f64(Q, K, V, rows), V, "2 loopback ranks, streamed shards")


@pytest.mark.parametrize("m,n,d,prec", [(32768, 65536, 128, "f32"), (32768, 65536, 512, "bf16")])
def test_streamed_batch_in_two_half_row_launches_hides_the_first_halfs_egress(m, n, d, prec, engine, O):
    """Round 6, behind $SDPA_DEBUG=two_wave=1: a one-batch call whose launch is long enough runs its streamed batch
    as TWO launches of half the rows each (each with the split count that fills the chip for its rows); the first half's merge and
    finish kernels sit between them on the compute stream and its rows cross PCIe and are widened under the second half's MFMAs.
    Bit-identical to the device-level launches of the two halves; the default one launch differs only by the summation order of its
    fewer splits (within the path's tolerance).  Built, measured, off by default: the tail shrinks, the launches lose more."""
    precision = None if prec == "f32" else "bf16"
    Q, K, V = O.make_inputs(m, n, d, d, "D1", seed=m + d)
    pkg = engine()
    assert stream_halves(pkg, m, n, d, d, prec)[0] == 1              # off by default: it did not pay (profiles/r06/two_wave_egress_ab.log)
    pkg = engine(SDPA_DEBUG="two_wave=1")
    assert stream_halves(pkg, m, n, d, d, prec)[0] == 2
    got = pkg.attention(Q, K, V, precision=precision)
    t = pkg.last_timing()
    assert t["streamed"] == 1 and t["fused_launches"] == 2, t
    assert np.array_equal(got, device_level(pkg, Q, K, V, 32768, precision=prec))
    assert np.array_equal(pkg.attention(Q, K, V, precision=precision), got)
    tol = fp32_tol(V) if prec == "f32" else bf16_tol(V)
    rows = np.arange(0, m, 331)
    check(got[rows], O.numpy_attention_f64(Q, K, V, rows), V, "two half-row launches", tol)
    pkg = engine()
    one = pkg.attention(Q, K, V, precision=precision)
    assert pkg.last_timing()["streamed"] == 1 and pkg.last_timing()["fused_launches"] == 1
    assert np.abs(one - got).max() <= 2 * tol


def test_streamed_launch_that_loses_a_ready_word_falls_back_to_the_chunked_schedule(engine, O):
    """A ready word that never comes (here: dropped on purpose, $SDPA_DEBUG=stream_drop_word) ends the launch -- every wait in the
    kernel watches the wall clock, and after the first timeout the launch's abort word ends all the others at once: ONE timeout for the
    call, not one per wave chain -- and the call is RE-RUN on the launch-per-chunk schedule: the caller gets the chunked schedule's result
    bit for bit and a line on stderr, never an error (the reference's attention() has no failure mode, attention-mpi.c:191-407).  The
    engine keeps that schedule from then on (no second timeout); a new engine probes again."""
    import time
    Q, K, V = O.make_inputs(8192, 32768, 128, 128, "D1", seed=5)               # two K/V groups (config 2 is ONE group)
    pkg = engine(SDPA_STREAMED=0)
    want = pkg.attention(Q, K, V)
    pkg = engine(SDPA_STREAM_TIMEOUT_MS=200, SDPA_DEBUG="stream_drop_word=2")  # the second K/V group is never announced
    assert len(pkg.plan(8192, 32768, 128, 128, 0, 1)["r"][0]["stream"]["end_tile"]) >= 2
    t0 = time.perf_counter()
    got = pkg.attention(Q, K, V)
    assert time.perf_counter() - t0 < 2.0, "one timeout (200 ms) + the chunked re-run"
    t = pkg.last_timing()
    assert t["streamed"] == 0 and t["fused_launches"] > 1, t
    assert np.array_equal(got, want)
    t0 = time.perf_counter()
    assert np.array_equal(pkg.attention(Q, K, V), want)                        # sticky: no timeout is sat out again
    assert time.perf_counter() - t0 < 0.15 and pkg.last_timing()["streamed"] == 0
    assert pkg.plan(8192, 32768, 128, 128, 0, 1)["r"][0]["stream"]["on"] == 0
    pkg = engine()
    got = pkg.attention(Q, K, V)                                               # a new engine streams again
    assert pkg.last_timing()["streamed"] == 1 and np.isfinite(got).all()


def test_prepare_probes_the_streamed_launch_and_disables_it_when_it_cannot_be_fed(engine, O):
    """sdpa_prepare()'s warm-up call IS the start-up probe: a streamed 8192 x 8192 call with a short wait bound.  When its words do not
    arrive ($SDPA_DEBUG drops group 0's) prepare still succeeds, quickly, says so once on stderr, and the timed call that follows runs
    the launch-per-chunk schedule -- correct, with no timeout inside it."""
    import time
    m, n, d = 8192, 32768, 128
    Q, K, V = O.make_inputs(m, n, d, d, "D2", seed=6)
    pkg = engine(SDPA_STREAMED=0)
    want = pkg.attention(Q, K, V)
    pkg = engine(SDPA_DEBUG="stream_drop_word=1", SDPA_PREPARE_WARM_MS=5)
    t0 = time.perf_counter()
    pkg.prepare(m, n, d, d)
    assert time.perf_counter() - t0 < 3.0
    t0 = time.perf_counter()
    got = pkg.attention(Q, K, V)
    dt = time.perf_counter() - t0
    assert pkg.last_timing()["streamed"] == 0 and dt < 0.15, (pkg.last_timing(), dt)
    assert np.array_equal(got, want)
    pkg = engine(SDPA_PREPARE_WARM_MS=5)                                       # the healthy case: the probe passes, the call streams
    pkg.prepare(m, n, d, d)
    got = pkg.attention(Q, K, V)
    assert pkg.last_timing()["streamed"] == 1
    rows = np.arange(0, m, 257)
    check(got[rows], O.numpy_attention_f64(Q, K, V, rows), V, "streamed after a passed probe")


def test_cli_prints_correct_when_the_streamed_probe_fails(O, tmp_path):
    """the one-shot CLI (sdpa_prepare + ONE timed call) on a runtime that cannot feed the streamed launch: `Correct!`, exit 0, the
    fallback announced on stderr"""
    import subprocess
    from conftest import PKG, ROOT
    cli = os.path.join(ROOT, PKG, "bin", "attention-hip")
    m, n, d = 8192, 16384, 128
    Q, K, V = O.make_inputs(m, n, d, d, "D1", seed=8)
    path = str(tmp_path / "probe.bin")
    O.write_case(path, Q, K, V, O.numpy_attention_f64(Q, K, V))
    r = subprocess.run([cli, path], capture_output=True, text=True, env=dict(os.environ, SDPA_VERBOSE="1", SDPA_DEBUG="stream_drop_word=1"))
    assert r.returncode == 0 and r.stdout.startswith("Correct!\nElapsed time: "), (r.stdout, r.stderr)
    assert "launch-per-chunk schedule" in r.stderr, r.stderr
    r = subprocess.run([cli, path], capture_output=True, text=True, env=dict(os.environ, SDPA_VERBOSE="1", HSA_ENABLE_SDMA="0"))
    assert r.returncode == 0 and r.stdout.startswith("Correct!\nElapsed time: "), (r.stdout, r.stderr)
    assert "one launch per K/V chunk" in r.stderr and "streamed (one persistent launch)" not in r.stderr, r.stderr      # no copy engines: never tried


# ------------------------------------------------- ... and its bf16 form (the tandem kernel's shapes: dv > 256) ------------------------------
def bf16_tol(V):
    return 1e-2 * max(1.0, float(np.abs(V).max()))


@pytest.mark.parametrize("m,n,dk,dv,dist,env", [
    (8192, 16384, 512, 512, "D1", {}),                        # BASELINE config 5's dims: 64 query blocks x 4 splits, several groups
    (32768, 8192, 512, 512, "D2", {}),                        # a whole-chip grid of 256 query blocks, ONE split
    (4096, 20011, 300, 400, "D3", {}),                        # padded dims (dk 512, 112 zero rows in the Vt image), ragged last tile
    (16384, 9001, 128, 512, "D4", {}),                        # the dk = 128 instantiation; adversarial spike late in the last group
    (20000, 16384, 256, 1024, "D2", {"SDPA_QBATCH": 8192}),   # dv = 1024: two 512-column chunks per query block; 3 batches
    (8192, 40000, 64, 320, "D2", {"SDPA_STREAM_CHUNK_MIN": 1024, "SDPA_KV_CHUNK_MAX": 4096}),   # many small groups
])
def test_streamed_bf16_first_batch_is_the_device_level_launch_bit_for_bit(m, n, dk, dv, dist, env, engine, orc, O):
    """The bf16 form of the streamed first batch (round 5): fused_bf16_tandem_stream_kernel is the tandem kernel's text with waits in
    front of the Q fragment load and of the LDS-DMA requests that cross into a new K/V group.  K travels as bf16 rows, V as column
    ranges of the Vt IMAGE the host writes (sdpa_host_cvt_vt's transposing converter on the pool) carried by pitched copies.  Same
    work, same order as sdpa_dev_shard_partial_bf16 on resident images (device converters): the same result bit for bit -- which
    also says the host's Vt image IS the device converter's; deterministic from call to call; the launch-per-chunk schedule agrees
    within the path's tolerance; both against the fp64 oracle."""
    Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=m + n)
    batch = int(env.get("SDPA_QBATCH", 32768))
    pkg = engine(**env)
    got = pkg.attention(Q, K, V, precision="bf16")
    t = pkg.last_timing()
    assert t["streamed"] == 1 and t["host_convert_threads"] > 0, t
    assert "fused_bf16_tandem" in t["last_kernel"], t["last_kernel"]
    if m <= batch:
        assert t["fused_launches"] == stream_halves(pkg, m, n, dk, dv, "bf16")[0] and t["last_kernel"].startswith("sdpa::fused_bf16_tandem_stream_kernel<"), t
    want = device_level(pkg, Q, K, V, batch, precision="bf16")
    assert np.array_equal(got, want), "streamed bf16 launch differs from the device-level launch in %d values (max %.3e)" % (
        (got != want).sum(), np.abs(got - want).max())
    for rep in range(2):
        assert np.array_equal(pkg.attention(Q, K, V, precision="bf16"), got), "call %d differs" % rep
    rows = np.sort(np.random.default_rng(m).choice(m, 48, replace=False))
    ref = O.numpy_attention_f64(Q, K, V, rows)
    check(got[rows], ref, V, "streamed bf16", bf16_tol(V))
    pkg = engine(SDPA_STREAMED=0, **env)
    old = pkg.attention(Q, K, V, precision="bf16")
    assert pkg.last_timing()["streamed"] == 0 and pkg.last_timing()["fused_launches"] > 1
    check(old[rows], ref, V, "launch per chunk, bf16", bf16_tol(V))
    assert np.abs(old - got).max() <= 2 * bf16_tol(V)


def test_streamed_bf16_calls_back_to_back_on_different_inputs_and_the_timeout(engine, O):
    """two input sets alternating in one engine (the images of a call land in the SAME device buffers as the previous call's while the
    launch is resident): every call equals its own set's first result bit for bit; and a ready word that never comes ends the launch
    inside $SDPA_STREAM_TIMEOUT_MS and the call falls back to the launch-per-chunk schedule"""
    import time
    m, n, d = 8192, 16384, 512
    sets = [O.make_inputs(m, n, d, d, dist, seed=seed) for dist, seed in (("D1", 21), ("D2", 22))]
    pkg = engine()
    rows = np.arange(0, m, max(1, m // 40))
    first = []
    for Q, K, V in sets:
        got = pkg.attention(Q, K, V, precision="bf16")
        assert pkg.last_timing()["streamed"] == 1, pkg.last_timing()
        check(got[rows], O.numpy_attention_f64(Q, K, V, rows), V, "first call of a set", bf16_tol(V))
        first.append(got)
    assert np.abs(first[0] - first[1]).max() > 1e-3
    for rep in range(4):
        for (Q, K, V), want in zip(sets, first):
            got = pkg.attention(Q, K, V, precision="bf16")
            assert np.array_equal(got, want), "round %d: %d values differ (max %.3e)" % (rep, (got != want).sum(), np.abs(got - want).max())
    Q, K, V = sets[0]
    pkg = engine(SDPA_STREAMED=0)
    chunked = pkg.attention(Q, K, V, precision="bf16")
    pkg = engine(SDPA_STREAM_TIMEOUT_MS=200, SDPA_DEBUG="stream_drop_word=2")
    assert len(pkg.plan(m, n, d, d, 2, 1)["r"][0]["stream"]["end_tile"]) >= 2
    t0 = time.perf_counter()
    got = pkg.attention(Q, K, V, precision="bf16")          # falls back: the chunked schedule's result, no error
    assert time.perf_counter() - t0 < 2.0
    assert pkg.last_timing()["streamed"] == 0 and np.array_equal(got, chunked)
    pkg = engine()
    got = pkg.attention(Q, K, V, precision="bf16")
    assert pkg.last_timing()["streamed"] == 1 and np.array_equal(got, first[0])


@pytest.mark.parametrize("P,m,n,dk,dv", [(2, 8192, 32768, 512, 512), (3, 8192, 40000, 300, 512)])
def test_streamed_bf16_shards_on_loopback_ranks(P, m, n, dk, dv, engine, O):
    """every rank of a K/V-sharded call streams ITS shard (its own packed Vt image in the staging, its own ready words); the merged
    result agrees with the fp64 oracle and with the launch-per-chunk schedule; page-locked caller arrays take the same path"""
    Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=P)
    pkg = engine(SDPA_VIRTUAL_GPUS=P)
    got = pkg.attention(Q, K, V, precision="bf16")
    t = pkg.last_timing()
    assert t["streamed"] == 1 and t["n_gpus"] == P, t
    rows = np.arange(0, m, 97)
    ref = O.numpy_attention_f64(Q, K, V, rows)
    check(got[rows], ref, V, "%d loopback ranks, streamed bf16 shards" % P, bf16_tol(V))
    assert np.array_equal(pkg.attention(Q, K, V, precision="bf16"), got)
    pkg = engine(SDPA_VIRTUAL_GPUS=P, SDPA_STREAMED=0)
    old = pkg.attention(Q, K, V, precision="bf16")
    assert pkg.last_timing()["streamed"] == 0
    assert np.abs(old - got).max() <= 2 * bf16_tol(V)


def test_converter_pool_on_the_numa_node_of_the_source_arrays_is_opt_in(engine, O):
    """$SDPA_DEBUG=host_cvt_pin=1 confines the converter pool's threads to the NUMA node the call's fp64 arrays live on (sampled pages; numpy
    arrays written by this thread live on ONE node) -- opt-in, because it did not pay (profiles/r05/converter_pool_numa_pin_ab.log);
    by default the threads run where the scheduler puts them.  Same bytes either way."""
    import glob
    nodes = len(glob.glob("/sys/devices/system/node/node[0-9]*"))
    Q, K, V = O.make_inputs(8192, 32768, 128, 128, "D1", seed=9)
    pkg = engine()
    a = pkg.attention(Q, K, V)
    t = pkg.last_timing()
    assert t["host_convert_threads"] > 0 and t["host_convert_node"] == -1, t
    pkg = engine(SDPA_HOST_CVT_PIN=1)
    b = pkg.attention(Q, K, V)
    t = pkg.last_timing()
    # (a node when every sampled page of Q, K and V lives on the same one; -1 when they do not -- the arrays were written by a thread the
    #  scheduler may have moved between them -- or when the host has one node)
    assert -1 <= t["host_convert_node"] < max(nodes, 1), t
    if nodes <= 1:
        assert t["host_convert_node"] == -1, t
    assert np.array_equal(a, b)
