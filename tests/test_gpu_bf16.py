"""GPU (-m gpu): the bf16-input MFMA path (BASELINE config 5) against the oracle.

Tolerance: max|got - fp64 oracle| <= 1e-2 * max(1, max|V|).  A tighter
check isolates the kernel from the input rounding: against the fp64 oracle evaluated on the
SAME bf16-rounded Q, K, V the budget is 4e-3 * max(1, max|V|) (P is rounded to bf16 once)."""
import os

import numpy as np
import pytest
import torch

from conftest import fp32_tol

pytestmark = pytest.mark.gpu


def bf16_tol(V):
    return 1e-2 * max(1.0, float(np.abs(V).max()))


def q_image_f64(Q):
    """what the kernels' Q operand means, in fp64: the Q image is bf16(Q * log2(e)/sqrtf(dk)) -- ONE
    rounding, taken after the softmax scale and the exp2 change of base are folded in
    (sdpa_dev_cvt_d2bf_q) -- so the Q the scores are computed from is that image divided by c again"""
    c = np.float32(1.44269504088896340736) * (np.float32(1.0) / np.sqrt(np.float32(Q.shape[1])))
    return to_bf16_f64(np.asarray(Q * np.float64(c), dtype=np.float32).astype(np.float64)) / np.float64(c)


def to_bf16_f64(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).to(torch.float64).numpy()


@pytest.fixture(scope="module")
def be(pkg):
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return pkg.HipBackend("cuda:0")


def dev_attention_bf16(pkg, be, Q, K, V):
    m, dk = Q.shape
    n, dv = V.shape
    sa = pkg.ShardedAttention(be, precision="bf16")
    sa.load_kv_from_root(K, V, n, dk, dv)
    qb = sa.convert_q(torch.from_numpy(np.ascontiguousarray(Q)).cuda())
    contrib, lmax, lsum = sa.batch_partial(qb)
    return be.finish_f64(contrib, lsum, dv).cpu().numpy()


def test_bf16_converts(be, pkg):
    x = torch.randn(70, 72, dtype=torch.float64, device="cuda") * 3
    y = be.cvt_d2bf(x)
    assert y.shape == (70, 128) and y.dtype == torch.bfloat16
    assert torch.equal(y[:, :72], x.to(torch.float32).to(torch.bfloat16)) and torch.all(y[:, 72:] == 0)
    # the Q image: bf16(Q * log2(e)/sqrtf(dk)), one rounding from the fp64 product
    q = be.cvt_d2bf_q(x)
    c = np.float32(1.44269504088896340736) * (np.float32(1.0) / np.sqrt(np.float32(72)))
    assert q.shape == (70, 128) and torch.all(q[:, 72:] == 0)
    assert torch.equal(q[:, :72], (x * float(c)).to(torch.float32).to(torch.bfloat16))
    v = torch.randn(100, 40, dtype=torch.float64, device="cuda")
    vt = be.cvt_d2bf_t(v)
    assert vt.shape == (64, 128)
    # key j of a row sits at kvpos(j): bits 2 and 3 of j swapped (include/sdpa_hip.h)
    pos = torch.tensor([be.lib.sdpa_dev_bf16_kvpos(j) for j in range(128)], device="cuda")
    assert sorted(pos.tolist()) == list(range(128))
    assert pos[:16].tolist() == [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]
    want = torch.zeros(64, 128, dtype=torch.bfloat16, device="cuda")
    want[:40, pos[:100]] = v.to(torch.float32).to(torch.bfloat16).t()
    assert torch.equal(vt, want)
    # a dv <= 256 shape's K image is plain rows
    k = torch.randn(100, 72, dtype=torch.float64, device="cuda")
    assert be.lib.sdpa_dev_bf16_tiled(40) == 0 and be.lib.sdpa_dev_bf16_tiled(256) == 0 and be.lib.sdpa_dev_bf16_tiled(257) == 1
    kb = be.cvt_d2bf_k(k, 40)
    assert kb.shape == (128, 128) and torch.equal(kb[:100, :72], k.to(torch.float32).to(torch.bfloat16)) and torch.all(kb[:100, 72:] == 0)


def tiled_images_reference(K, V, dk_pad, dv_pad):
    """the TILED images (include/sdpa_hip.h, dv > 256) built in numpy from the definition: uint16 bit patterns"""
    def bits(x):
        return torch.from_numpy(np.ascontiguousarray(x)).to(torch.float32).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    n, dk = K.shape
    dv = V.shape[1]
    npad = (n + 31) // 32 * 32
    kb = np.zeros((npad, dk_pad), np.uint16)
    kb[:n, :dk] = bits(K)
    swz = min(15, dk_pad // 8 - 1)
    kimg = np.zeros_like(kb)
    for r in range(npad):
        row = kb[r].reshape(-1, 8)
        kimg[r] = row[np.arange(dk_pad // 8) ^ (r & swz)].reshape(-1)          # chunk position p holds chunk p ^ (r & swz)
    vb = np.zeros((npad, dv_pad), np.uint16)
    vb[:n, :dv] = bits(V)
    kvpos = lambda j: (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1)
    vimg = np.zeros((npad // 32, dv_pad // 512, 512, 32), np.uint16)
    for t in range(npad // 32):
        tile = vb[32 * t:32 * t + 32]                                            # [key, column]
        line = np.zeros((dv_pad, 32), np.uint16)
        for j in range(32):
            line[:, kvpos(j)] = tile[j]
        line = line.reshape(dv_pad, 4, 8)
        col = np.arange(dv_pad)
        x = (col >> 2) & 3
        out = np.empty_like(line)
        for q in range(4):
            out[col, q] = line[col, q ^ x]                                       # chunk position q holds chunk q ^ x
        vimg[t] = out.reshape(dv_pad // 512, 512, 32)
    return kimg, vimg.reshape(-1)


@pytest.mark.parametrize("n,dk,dv", [(100, 72, 300), (4096, 512, 512), (77, 512, 700), (33, 64, 260)])
def test_bf16_tiled_images_are_the_documented_layout(n, dk, dv, be):
    rng = np.random.default_rng(n + dk)
    K, V = rng.standard_normal((n, dk)), rng.standard_normal((n, dv))
    kb = be.cvt_d2bf_k(torch.from_numpy(K).cuda(), dv)
    vt = be.cvt_d2bf_t(torch.from_numpy(V).cuda())
    dkp, dvp = be.lib.sdpa_dev_bf16_ld(dk), be.lib.sdpa_dev_bf16_dvp(dv)
    kimg, vimg = tiled_images_reference(K, V, dkp, dvp)
    assert kb.shape == kimg.shape and np.array_equal(kb.view(torch.int16).cpu().numpy().view(np.uint16), kimg)
    assert vt.numel() == vimg.size and np.array_equal(vt.view(torch.int16).cpu().numpy().view(np.uint16).reshape(-1), vimg)


SHAPES = [
    # m,    n,   dk,  dv, dist
    (64,   64,   64,  64, "D1"),
    (1,     1,    1,   1, "D2"),
    (33,   33,   33,  33, "D2"),
    (200,   5,   16,  16, "D2"),      # n < tile
    (130, 333,  128, 128, "D2"),
    (100, 257,   72,  40, "D2"),      # dims padded to 128 / 64
    (96,  1000,  64,  64, "D4"),      # late spike key
    (257, 2048, 128, 128, "D3"),      # peaky
    (64,  300,  256, 256, "D1"),
    (48,  300,  512, 512, "D1"),      # BASELINE config 5 dims: two dv chunks of 256
    (40,  200,  100, 200, "D2"),      # dv padded to 256, dk to 128
    (32,   96,  512,  64, "D1"),
    (129, 700,  300, 700, "D1"),      # dk -> 512, dv -> 2 chunks of 512
    (96,  1000,  64, 300, "D4"),      # tandem kernel (dv > 256), dk padded to 64, late spike key
    (64,  2048, 128, 512, "D3"),      # tandem, dk 128, peaky
    (70,  333,  256, 384, "D2"),      # tandem, dk 256, ragged last tile
    (300, 4096, 512, 512, "D2"),      # tandem with in-GPU K/V splits
    (200,   5,  512, 512, "D2"),      # tandem, n < tile
    (33,   64,  400, 257, "D1"),      # tandem, both dims padded
]


@pytest.mark.parametrize("m,n,dk,dv,dist", SHAPES)
def test_bf16_shapes(m, n, dk, dv, dist, pkg, be, orc, O):
    Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=m + n + 1)
    got = dev_attention_bf16(pkg, be, Q, K, V)
    assert got.shape == (m, dv) and np.isfinite(got).all()
    want = orc.attention_f64(Q, K, V)
    assert np.abs(got - want).max() <= bf16_tol(V)
    same_inputs = orc.attention_f64(q_image_f64(Q), to_bf16_f64(K), to_bf16_f64(V))
    assert np.abs(got - same_inputs).max() <= 4e-3 * max(1.0, np.abs(V).max())


def test_bf16_tandem_steep_ramp_takes_the_redo_pass(pkg, be, O):
    """dv > 256 runs the tandem kernel, which has no accumulator rescale and the reference exponent
    zero: a q block with a row whose sum of 2^score leaves [2^-80, 2^80] is flagged and redone by the
    general kernel (which reads the same TILED images).
    Block 0 (rows 0..127) climbs 0.5 nat per key, block 1 has flat scores and must stay on the
    tandem kernel's own result."""
    m, n, d = 256, 1024, 512
    rng = np.random.default_rng(11)
    Q = np.zeros((m, d)); K = np.zeros((n, d))
    Q[:128, 0] = np.sqrt(d)                       # score_j = K[j, 0]
    K[:, 0] = 0.5 * np.arange(n)                  # exact in bf16 up to 128, rounded above: same inputs both sides
    K[:, 1:] = rng.standard_normal((n, d - 1)) * 0.1
    V = rng.standard_normal((n, d))
    got = dev_attention_bf16(pkg, be, Q, K, V)
    want = O.numpy_attention_f64(q_image_f64(Q), to_bf16_f64(K), to_bf16_f64(V))
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 4e-3 * max(1.0, np.abs(V).max())
    # flat block: plain mean of V
    assert np.abs(got[128:] - to_bf16_f64(V).mean(axis=0)).max() <= 4e-3 * max(1.0, np.abs(V).max())
    # with K/V splits on top (few q blocks, long n)
    n2 = 8192
    K2 = np.zeros((n2, d)); K2[:, 0] = 0.25 * np.arange(n2); K2[:, 1:] = rng.standard_normal((n2, d - 1)) * 0.1
    V2 = rng.standard_normal((n2, d))
    assert pkg.load().sdpa_dev_kv_splits_bf16(m, n2, d, d) > 1
    got2 = dev_attention_bf16(pkg, be, Q, K2, V2)
    want2 = O.numpy_attention_f64(q_image_f64(Q), to_bf16_f64(K2), to_bf16_f64(V2))
    assert np.isfinite(got2).all() and np.abs(got2 - want2).max() <= 4e-3 * max(1.0, np.abs(V2).max())


@pytest.mark.parametrize("d", [64, 128, 256])
def test_bf16_duo_steep_scores_take_the_redo_pass(d, pkg, be, O):
    """dk, dv <= 256 run the duo kernel (two query blocks per wave, 256-row workgroups), which like
    the tandem kernel computes against the reference exponent zero: a workgroup in which some row's sum
    of 2^score leaves [2^-80, 2^80] flags its two 128-row blocks and the general kernel redoes them.
    Rows 0..127 climb 0.5 nat per key (workgroup 0 is redone, including its flat rows 128..255);
    workgroup 1 (rows 256..511, flat scores) must stay on the duo kernel's own result."""
    m, n = 512, 1024
    rng = np.random.default_rng(17 + d)
    Q = np.zeros((m, d)); K = np.zeros((n, d))
    Q[:128, 0] = np.sqrt(d)                       # score_j = K[j, 0]
    K[:, 0] = 0.5 * np.arange(n)
    K[:, 1:] = rng.standard_normal((n, d - 1)) * 0.1
    V = rng.standard_normal((n, d))
    got = dev_attention_bf16(pkg, be, Q, K, V)
    want = O.numpy_attention_f64(q_image_f64(Q), to_bf16_f64(K), to_bf16_f64(V))
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 4e-3 * max(1.0, np.abs(V).max())
    assert np.abs(got[128:] - to_bf16_f64(V).mean(axis=0)).max() <= 4e-3 * max(1.0, np.abs(V).max())
    # with K/V splits on top, and a ragged last tile
    n2 = 8192 + 13
    K2 = np.zeros((n2, d)); K2[:, 0] = 0.25 * np.arange(n2); K2[:, 1:] = rng.standard_normal((n2, d - 1)) * 0.1
    V2 = rng.standard_normal((n2, d))
    assert pkg.load().sdpa_dev_kv_splits_bf16(m, n2, d, d) > 1
    got2 = dev_attention_bf16(pkg, be, Q, K2, V2)
    want2 = O.numpy_attention_f64(q_image_f64(Q), to_bf16_f64(K2), to_bf16_f64(V2))
    assert np.isfinite(got2).all() and np.abs(got2 - want2).max() <= 4e-3 * max(1.0, np.abs(V2).max())


@pytest.mark.parametrize("dk,dv", [(64, 64), (64, 128), (128, 64), (128, 128), (64, 256), (256, 64), (128, 256),
                                   (256, 128), (256, 256), (72, 200), (200, 40)])
def test_bf16_duo_every_instantiation(dk, dv, pkg, be, orc, O):
    """each <DK, DV> instantiation of the duo kernel at a size with several 256-row workgroups,
    in-GPU K/V splits and ragged rows / keys, vs the fp64 oracle, plus run-to-run bit identity
    (its tiles move by LDS-DMA behind counted waits: a missing wait shows as rare wrong tiles)"""
    m, n = 700, 5000 + dk
    Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=dk + dv)
    got = dev_attention_bf16(pkg, be, Q, K, V)
    assert np.isfinite(got).all()
    assert np.abs(got - orc.attention_f64(Q, K, V)).max() <= bf16_tol(V)
    for _ in range(5):
        assert np.array_equal(dev_attention_bf16(pkg, be, Q, K, V), got)


@pytest.mark.parametrize("d", [128, 512])
def test_bf16_kv_splits_and_triple(d, pkg, be, O):
    """long K/V with few query blocks: in-GPU splits.  The triple of the fixed-reference kernels
    (d = 128: duo, d = 512: tandem) is relative to a power of two, not to the row max: lmax is the
    reference exponent that puts lsum in [1, 2), and lmax + ln(lsum) is the row's log-sum-exp of the
    scores of the bf16 operand images"""
    m, n = 256, 8192
    assert pkg.load().sdpa_dev_kv_splits_bf16(m, n, d, d) > 1
    Q, K, V = O.make_inputs(m, n, d, d, "D2", seed=4)
    got = dev_attention_bf16(pkg, be, Q, K, V)
    assert np.abs(got - O.numpy_attention_f64(Q, K, V)).max() <= bf16_tol(V)
    sa = pkg.ShardedAttention(be, precision="bf16")
    sa.load_kv_from_root(K, V, n, d, d)
    _, lmax, lsum = sa.batch_partial(sa.convert_q(torch.from_numpy(Q).cuda()))
    lmax = lmax.cpu().numpy().astype(np.float64); lsum = lsum.cpu().numpy().astype(np.float64)
    s = (q_image_f64(Q) @ to_bf16_f64(K).T) / np.sqrt(np.float32(d))
    lse = s.max(axis=1) + np.log(np.exp(s - s.max(axis=1, keepdims=True)).sum(axis=1))
    assert np.abs(lmax + np.log(lsum) - lse).max() <= 1e-4 * max(1.0, np.abs(lse).max())
    assert (lsum > 0).all() and np.abs(lmax - s.max(axis=1)).max() <= np.log(n) + 1.0


@pytest.mark.parametrize("d", [128, 512])
@pytest.mark.parametrize("offset", [-40.0, 40.0, -70.0, 70.0])
def test_bf16_fixed_reference_range(d, offset, pkg, be, O):
    """The duo and tandem kernels compute P = 2^score against the reference exponent ZERO.  Rows whose
    scores sit +-40 nats from zero are still inside their range (row sums within 2^+-80); at +-70 nats
    (2^+-101) the row sum leaves it -- overflow on one side, P flushed towards zero on the other -- and
    the general kernel redoes the block.  Either way the answer is the oracle's."""
    m, n = 256, 2048 + 5
    rng = np.random.default_rng(int(1000 + d + offset))
    Q = rng.standard_normal((m, d)) * 0.3
    K = rng.standard_normal((n, d)) * 0.3
    Q[:, 0] = np.sqrt(d)
    K[:, 0] = offset                              # every score of every row moves by `offset` nats
    V = rng.standard_normal((n, d))
    got = dev_attention_bf16(pkg, be, Q, K, V)
    want = O.numpy_attention_f64(q_image_f64(Q), to_bf16_f64(K), to_bf16_f64(V))
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 4e-3 * max(1.0, np.abs(V).max())


def test_bf16_host_level_flag_and_shard_merge(pkg, be, orc, O):
    Q, K, V = O.make_inputs(300, 900, 128, 128, "D2", seed=2)
    want = orc.attention_f64(Q, K, V)
    got = pkg.attention(Q, K, V, precision="bf16")
    assert np.isfinite(got).all() and np.abs(got - want).max() <= bf16_tol(V)
    # it really took the bf16 path: the fp32 path is ~100x closer
    f32 = pkg.attention(Q, K, V)
    assert np.abs(f32 - want).max() <= fp32_tol(V) < np.abs(got - want).max()
    # 3 shards (one after the other on this GPU) merged with the merge kernels == unsharded
    qb = be.cvt_d2bf_q(torch.from_numpy(Q).cuda())
    triples = []
    for r in range(3):
        c, d0 = pkg.owner_count(900, 3, r), pkg.owner_disp(900, 3, r)
        sa = pkg.ShardedAttention(be, precision="bf16")
        sa.load_kv_from_root(K[d0:d0 + c], V[d0:d0 + c], c, 128, 128)
        triples.append(sa.batch_partial(qb))
    gmax = torch.stack([t[1] for t in triples]).max(dim=0).values
    for contrib, lmax, lsum in triples:
        be.merge_rescale(contrib, lsum, lmax, gmax, 128)
    gsum = torch.stack([t[2] for t in triples]).sum(dim=0)
    for contrib, _, _ in triples:
        be.merge_normalise(contrib, gsum, 128)
    merged = be.cvt_f2d(torch.stack([t[0] for t in triples]).sum(dim=0), 128).cpu().numpy()
    assert np.abs(merged - want).max() <= bf16_tol(V)


def test_bf16_empty_shard(pkg, be):
    qb = torch.zeros(40, 64, dtype=torch.bfloat16, device="cuda")
    contrib, lmax, lsum = be.shard_partial_bf16(qb, None, None, 0, 64, 64)
    assert torch.all(contrib[:, :64] == 0) and torch.all(lsum == 0) and torch.all(torch.isneginf(lmax))


def test_bf16_config5_rows(pkg, O):
    """BASELINE config 5 at a quarter of m (m=8192, n=65536, dk=dv=512) through the host-level
    boundary: a row subset against the fp64 oracle"""
    m, n, d = 8192, 65536, 512
    rng = np.random.default_rng(5)
    Q, K, V = (rng.uniform(-1, 1, s) for s in ((m, d), (n, d), (n, d)))
    got = pkg.attention(Q, K, V, precision="bf16")
    assert np.isfinite(got).all()
    rows = rng.choice(m, 48, replace=False)
    assert np.abs(got[rows] - O.numpy_attention_f64(Q, K, V, rows)).max() <= bf16_tol(V)


def test_random_shape_sweep_bf16(pkg, be, orc, O):
    """30 seeded random shapes across every bf16 kernel.  Two checks per case:
    (1) the KERNEL: against the fp64 oracle evaluated on the very operands the kernel multiplies
        (the bf16 images of Q*log2e/sqrt(dk), K, V) -- 4e-3 * max(1, max|V|);
    (2) the PATH: against the fp64 oracle on the original inputs -- BASELINE.md's bf16 bar,
        1e-2 * max(1, max|V|).  Rounding the OPERANDS to bf16 already costs e_in = |oracle(images) -
        oracle(inputs)|; on peaky inputs with a handful of keys (score std 4, n = 2, dk = 512) e_in alone
        can sit above that bar, and no kernel can be closer to the inputs' answer than its operands
        are -- so where e_in nearly fills the bar the path is held to e_in + the kernel tolerance."""
    rng = np.random.default_rng(77)
    worst_kernel = worst_path = 0.0
    for it in range(30):
        m = int(rng.integers(1, 260))
        n = int(rng.integers(1, 600))
        dk = int(rng.choice([1, 8, 33, 64, 65, 128, 130, 256, 300, 512]))
        dv = int(rng.choice([1, 16, 40, 64, 100, 128, 200, 256, 257, 512]))
        dist = ["D1", "D2", "D3", "D4"][it % 4]
        Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=2000 + it)
        got = dev_attention_bf16(pkg, be, Q, K, V)
        want = orc.attention_f64(Q, K, V)
        on_images = orc.attention_f64(q_image_f64(Q), to_bf16_f64(K), to_bf16_f64(V))
        assert np.isfinite(got).all(), (m, n, dk, dv, dist)
        vmax = max(1.0, np.abs(V).max())
        e_kernel = np.abs(got - on_images).max()
        e_in = np.abs(on_images - want).max()
        e_path = np.abs(got - want).max()
        assert e_kernel <= 4e-3 * vmax, ("kernel", m, n, dk, dv, dist, e_kernel)
        assert e_path <= max(bf16_tol(V), e_in + 4e-3 * vmax), ("path", m, n, dk, dv, dist, e_path, e_in)
        worst_kernel = max(worst_kernel, e_kernel / (4e-3 * vmax))
        worst_path = max(worst_path, e_path / bf16_tol(V))
    print("worst kernel err / 4e-3*vmax: %.3f   worst path err / bf16 bar: %.3f" % (worst_kernel, worst_path))


def test_bf16_race_screen_repeatability(pkg, be, O):
    for (m, n, dk, dv) in [(384, 5000, 128, 128), (200, 2100, 512, 512)]:
        Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=78)
        sa = pkg.ShardedAttention(be, precision="bf16")
        sa.load_kv_from_root(K, V, n, dk, dv)
        qb = sa.convert_q(torch.from_numpy(Q).cuda())
        first = None
        for it in range(25):
            contrib, lmax, lsum = sa.batch_partial(qb)
            cur = (contrib[:, :dv].clone(), lmax.clone(), lsum.clone())
            if first is None:
                first = cur
            else:
                assert all(torch.equal(a_, b_) for a_, b_ in zip(cur, first)), "launch %d differs" % it


# ---------------------------------------------------------------- tandem kernel (dv > 256, round 3) --------
@pytest.mark.parametrize("m,n,dk,dv,dist", [
    (260, 5000, 512, 512, "D2"),       # BASELINE config 5's dims: ragged rows, in-GPU splits
    (128, 32, 512, 512, "D1"),         # one tile
    (129, 33, 512, 512, "D2"),         # two tiles, the second ragged; a ragged query block
    (200, 64, 256, 384, "D2"),         # two full tiles, dv padded to 512
    (300, 96, 128, 300, "D4"),         # three tiles
    (70, 700, 64, 260, "D2"),          # narrowest K rows (one DMA piece per wave)
    (140, 3000, 512, 700, "D2"),       # dv > 512: two chunks of 512 columns
    (513, 2048, 384, 512, "D3"),       # dk padded to 512, peaky scores
    (32, 4100, 512, 1024, "D1"),       # fewer rows than a pair holds
])
def test_bf16_tandem_kernel_against_the_oracle_and_itself(m, n, dk, dv, dist, pkg, be, orc, O):
    """dv > 256: the tandem kernel on the TILED images (two waves share 64 rows and split the columns, P handed over through
    LDS; round 6: one barrier per step, every DMA piece of a step issued behind it, MFMAs carried across it).  Its tiles move
    by LDS-DMA behind counted waits and ONE barrier: a missing wait shows as rare wrong tiles -- 6 launches must be IDENTICAL;
    the result within the kernel's budget of the fp64 oracle on the very operand images it multiplies, and within the bf16
    tolerance of the oracle on the inputs."""
    Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=m + n + dk)
    sa = pkg.ShardedAttention(be, precision="bf16")
    sa.load_kv_from_root(K, V, n, dk, dv)
    qb = sa.convert_q(torch.from_numpy(np.ascontiguousarray(Q)).cuda())
    want = tuple(t.clone() for t in sa.batch_partial(qb))
    assert "fused_bf16_tandem_kernel" in pkg.last_launch()["kernel"]
    for it in range(5):
        got = sa.batch_partial(qb)
        for name, g, w in zip(("contrib", "lmax", "lsum"), got, want):
            g, w = (g[:, :dv], w[:, :dv]) if g.dim() == 2 else (g, w)
            assert torch.equal(g, w), "launch %d: %s differs from the first launch" % (it, name)
    res = be.finish_f64(got[0], got[2], dv).cpu().numpy()
    assert np.isfinite(res).all()
    assert np.abs(res - orc.attention_f64(Q, K, V)).max() <= bf16_tol(V)
    same_inputs = orc.attention_f64(q_image_f64(Q), to_bf16_f64(K), to_bf16_f64(V))
    assert np.abs(res - same_inputs).max() <= 4e-3 * max(1.0, np.abs(V).max())


def test_bf16_tandem_steep_scores_take_the_redo_pass(pkg, be, O):
    """rows whose scores leave the fixed reference range are flagged by the wave that SCORES them and redone by
    the general kernel -- also when the other half of their columns sits in the partner wave"""
    m, n, d = 256, 2048, 512
    rng = np.random.default_rng(7)
    Q, K, V = (rng.standard_normal(s) for s in ((m, d), (n, d), (n, d)))
    Q[5] *= 40.0                       # one row of wave 0 ...
    Q[100] *= 40.0                     # ... and one of wave 3's: both pairs redo
    want = O.numpy_attention_f64(Q, K, V)
    got = dev_attention_bf16(pkg, be, Q, K, V)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= bf16_tol(V)


def test_bf16_path_on_the_mid_size_reference_golden(pkg, be, O):
    """tests/golden/mid_d512_D3 (m = 256, n = 8192, d = 512; answer block = output of the reference's own attention(),
    oracle/make_golden.py): the tandem kernel over 256 K/V tiles with in-GPU splits, against reference bytes at
    the bf16 path's tolerance."""
    import json
    case = [c for c in json.load(open(os.path.join(O.golden_dir(), "INDEX.json"))) if c["name"] == "mid_d512_D3"][0]
    Q, K, V, ans = O.load_golden(case)
    got = dev_attention_bf16(pkg, be, Q, K, V)
    assert np.isfinite(got).all()
    err = np.abs(got - ans).max()
    print("mid_d512_D3 bf16: max|err| %.3e (tol %.3e)" % (err, bf16_tol(V)))
    assert err <= bf16_tol(V)
