"""GPU: seeded differential fuzz of the whole path against the fp64 oracle -- shapes drawn on both sides
of every kernel's dispatch and tile boundary (head dims 1..1100, ragged rows and keys, key counts
around the chunk and split thresholds), all four input distributions, fp32 and bf16, the device-level
path and the host boundary with randomly drawn pipeline knobs.  $SDPA_FUZZ_CASES sets the number of
cases per test (default 60; tools/gpu_fuzz.sh runs 600)."""
import os

import numpy as np
import pytest
import torch

from conftest import knob_env

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("SDPA_FUZZ_CASES", "60"))
DK = [1, 3, 16, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 192, 255, 256, 257, 300, 383, 384, 385, 500, 512,
      513, 600, 768, 769, 1000, 1024, 1025, 1100]
DV = [1, 5, 31, 32, 33, 64, 65, 100, 128, 129, 200, 256, 257, 384, 512, 513, 700, 1024, 1500]
DENSE = [64, 128, 256]


def fp32_tol(V):
    return 5e-5 * max(1.0, float(np.abs(V).max()))


def bf16_tol(V):
    return 1e-2 * max(1.0, float(np.abs(V).max()))


def draw_shape(rng, bf16, it):
    if it % 3 == 0:                       # dense power-of-two head dims: the pipelined / duo / wide kernels
        dk = int(rng.choice(DENSE + ([512] if bf16 else [])))
        dv = int(rng.choice(DENSE + ([512] if bf16 else [])))
    else:
        dk = int(rng.choice([d for d in DK if not bf16 or d <= 512]))
        dv = int(rng.choice([d for d in DV if d <= 1024 or (not bf16 and dk <= 1024)]))   # wider only on the MFMA kernels
    m = int(rng.choice([1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513, 700]))
    big = dk * dv > 300 * 300
    n = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 255, 256, 257, 1000] +
                       ([] if big else [1023, 1024, 1025, 2047, 2049, 4096, 5000, 9000])))
    return m, n, dk, dv


@pytest.fixture(scope="module")
def be(pkg):
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return pkg.HipBackend("cuda:0")


def dev_attention(pkg, be, Q, K, V, prec):
    m, dk = Q.shape
    n, dv = V.shape
    sa = pkg.ShardedAttention(be, precision=prec or "f32")
    sa.load_kv_from_root(K, V, n, dk, dv)
    qf = sa.convert_q(torch.from_numpy(np.ascontiguousarray(Q)).cuda())
    contrib, lmax, lsum = sa.batch_partial(qf)
    return be.finish_f64(contrib, lsum, dv).cpu().numpy()


@pytest.mark.parametrize("prec", [None, "bf16"])
def test_fuzz_device_level(prec, pkg, be, orc, O):
    rng = np.random.default_rng(77 if prec else 55)
    worst = (0.0, None)
    for it in range(CASES):
        m, n, dk, dv = draw_shape(rng, prec == "bf16", it)
        # (bf16: no D3 -- on near-arg-max rows the operands' rounding alone can exceed the 1e-2 bar; the
        #  kernel-vs-operand-image error of such rows is bounded in tests/test_gpu_bf16.py's sweep)
        dist = (["D1", "D2", "D4"] if prec else ["D1", "D2", "D3", "D4"])[int(rng.integers(0, 3 if prec else 4))]
        Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=5000 + it)
        got = dev_attention(pkg, be, Q, K, V, prec)
        want = O.numpy_attention_f64(Q, K, V)
        tol = bf16_tol(V) if prec else fp32_tol(V)
        assert got.shape == want.shape and np.isfinite(got).all(), (m, n, dk, dv, dist, prec)
        rel = np.abs(got - want).max() / tol
        assert rel <= 1.0, "case %d %s: err/tol = %.3f" % (it, (m, n, dk, dv, dist, prec), rel)
        if rel > worst[0]:
            worst = (rel, (m, n, dk, dv, dist))
    print("worst err/tol %.3f at %s over %d cases" % (worst[0], worst[1], CASES))


@pytest.mark.parametrize("prec", [None, "bf16"])
def test_fuzz_host_boundary(prec, pkg, orc, O, monkeypatch):
    """sdpa_attention_f64 with numpy (pageable) arrays and randomly drawn schedule knobs: every exact
    schedule must give the oracle's answer"""
    rng = np.random.default_rng(99 if prec else 11)
    for it in range(max(10, CASES // 2)):
        m, n, dk, dv = draw_shape(rng, prec == "bf16", it)
        if it % 4 == 0:
            n = int(rng.choice([3000, 5000, 9000, 12000])) if dk * dv <= 256 * 256 else n
        dist = ["D1", "D2", "D4"][int(rng.integers(0, 3))]
        knobs = {"SDPA_KV_CHUNK_MIN": int(rng.choice([1024, 2048, 4096])),
                 "SDPA_KV_CHUNK_MAX": int(rng.choice([1024, 4096, 16384])),
                 "SDPA_QBATCH": int(rng.choice([64, 256, 32768])),
                 "SDPA_ROW_PIECES": int(rng.choice([1, 2, 4])),
                 "SDPA_PIECE_MIN_ROWS": int(rng.choice([128, 4096])),
                 "SDPA_PROGRESSIVE_PIN": int(rng.integers(0, 2)),
                 # convert placement (round 3): device, host threads, or chosen per problem
                 "SDPA_HOST_CVT": str(rng.choice(["0", "1", "auto"])),
                 "SDPA_HOST_CVT_THREADS": int(rng.choice([1, 5, 32]))}
        for k, v in knob_env(knobs).items():
            monkeypatch.setenv(k, v)
        Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=9000 + it)
        got = pkg.attention(Q, K, V, precision=prec, flags=int(rng.integers(0, 2)))
        want = O.numpy_attention_f64(Q, K, V)
        tol = bf16_tol(V) if prec else fp32_tol(V)
        assert got.shape == want.shape and np.isfinite(got).all(), (m, n, dk, dv, dist, prec, knobs)
        rel = np.abs(got - want).max() / tol
        assert rel <= 1.0, "case %d %s %s: err/tol = %.3f" % (it, (m, n, dk, dv, dist, prec), knobs, rel)


def test_fuzz_loopback_ranks(pkg, orc, O, monkeypatch):
    """the C host's P > 1 pipeline on loopback ranks (SDPA_VIRTUAL_GPUS): random P, plan, merge, precision,
    shapes with n < P (empty shards) and ragged batches"""
    assert torch.cuda.is_available()
    rng = np.random.default_rng(31337)
    knob_names = ("SDPA_VIRTUAL_GPUS", "SDPA_QBATCH", "SDPA_EGRESS", "SDPA_HOST_CVT", "SDPA_DEBUG")
    try:
        for it in range(max(8, CASES // 4)):
            P = int(rng.choice([2, 3, 4, 5, 8, 16]))
            prec = "bf16" if it % 3 == 2 else None
            m, n, dk, dv = draw_shape(rng, prec == "bf16", it)
            if it % 2 == 0 and dk * dv <= 256 * 256:
                n = int(rng.choice([P - 1, P, P + 1, 700, 3000, 5000]))
            n = max(1, n)
            plan = "qrows" if it % 5 == 4 else None
            merge = "allreduce" if it % 2 else None
            knobs = {"SDPA_VIRTUAL_GPUS": P, "SDPA_QBATCH": int(rng.choice([64, 192, 32768])),
                     "SDPA_KV_CHUNK_MIN": 1024, "SDPA_KV_CHUNK_MAX": int(rng.choice([1024, 4096])),
                     "SDPA_ROW_PIECES": int(rng.choice([1, 4])), "SDPA_PIECE_MIN_ROWS": 128,
                     # the round-3 schedule switches: every combination is an exact schedule
                     "SDPA_EGRESS": str(rng.choice(["root", "scatter"])), "SDPA_ENQUEUE_THREADS": int(rng.integers(0, 2)),
                     "SDPA_HOST_CVT": str(rng.choice(["0", "1"])), "SDPA_PROGRESSIVE_PIN": int(rng.integers(0, 2))}
            pkg.shutdown()
            for k, v in knob_env(knobs).items():
                monkeypatch.setenv(k, v)
            pkg.init(1)
            dist = ["D1", "D2", "D4"][int(rng.integers(0, 3))]
            Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=12000 + it)
            got = pkg.attention(Q, K, V, precision=prec, plan=plan, merge=merge)
            t = pkg.last_timing()
            assert t["n_gpus"] == P and t["virtual_ranks"] == 1, t
            want = O.numpy_attention_f64(Q, K, V)
            tol = bf16_tol(V) if prec else fp32_tol(V)
            assert np.isfinite(got).all(), (P, m, n, dk, dv, dist, prec, plan, merge)
            rel = np.abs(got - want).max() / tol
            assert rel <= 1.0, "case %d P=%d %s %s/%s: err/tol = %.3f" % (it, P, (m, n, dk, dv, dist, prec), plan, merge, rel)
    finally:
        pkg.shutdown()
        for k in knob_names:
            monkeypatch.delenv(k, raising=False)
        pkg.init(1)


@pytest.mark.parametrize("prec", [None, "bf16"])
@pytest.mark.parametrize("shape", [(300, 1000, 128, 128), (260, 700, 256, 256), (130, 5000, 64, 64), (140, 900, 100, 72),
                                   (70, 600, 512, 512)])
def test_nan_in_one_query_row_stays_in_that_row(prec, shape, pkg, be, O):
    """a NaN in Q poisons exactly its own output row (as in attention.c:28-75, rows are independent):
    no other row of the workgroup, wave or redo block may change, nothing hangs"""
    m, n, dk, dv = shape
    Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=m + dk)
    bad = [3, m - 1] if m > 130 else [3]
    Qn = Q.copy()
    Qn[bad, 0] = np.nan
    tol = bf16_tol(V) if prec else fp32_tol(V)
    want = O.numpy_attention_f64(Q, K, V)
    for got in (dev_attention(pkg, be, Qn, K, V, prec), pkg.attention(Qn, K, V, precision=prec)):
        assert np.isnan(got[bad]).all()
        good = np.ones(m, dtype=bool)
        good[bad] = False
        assert np.isfinite(got[good]).all()
        assert np.abs(got[good] - want[good]).max() <= tol
