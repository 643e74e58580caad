"""GPU (-m gpu): parity of the HIP hot path with the oracle, through the C ABI.

Tolerance (BASELINE.md section 4, fp32 compute / fp64 output): the whole result array,
max|got - fp64 oracle| <= 5e-5 * max(1, max|V|), any NaN/Inf fails."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT, fp32_tol

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, PKG, "bin", "attention-hip")


def golden_cases():
    return json.load(open(os.path.join(GOLD, "INDEX.json")))


@pytest.fixture(scope="module")
def be(pkg):
    assert torch.cuda.is_available(), "the -m gpu tests need a real MI355X"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName
    torch.cuda.set_device(0)
    return pkg.HipBackend("cuda:0")


def dev_attention(pkg, be, Q, K, V):
    """device-level path: cvt_d2f -> fused kernel (+ split merge) -> finish_f64"""
    m, dk = Q.shape
    n, dv = V.shape
    sa = pkg.ShardedAttention(be)
    sa.load_kv_from_root(K, V, n, dk, dv)
    qf = be.cvt_d2f(torch.from_numpy(np.ascontiguousarray(Q)).cuda())
    contrib, lmax, lsum = sa.batch_partial(qf)
    return be.finish_f64(contrib, lsum, dv).cpu().numpy()


def check(got, want, V, what="", tol=None):
    tol = fp32_tol(V) if tol is None else tol
    assert got.shape == want.shape
    assert np.isfinite(got).all(), what + ": non-finite values"
    err = np.abs(got - want).max()
    assert err <= tol, "%s: max|err| %.3e > %.3e" % (what, err, tol)
    return err


# ---------------------------------------------------------------- golden vectors -------------
@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_golden_host_level(case, pkg, O):
    Q, K, V, ans = O.load_golden(case)
    check(pkg.attention(Q, K, V), ans, V, "host level")


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_golden_device_level(case, pkg, be, O):
    Q, K, V, ans = O.load_golden(case)
    check(dev_attention(pkg, be, Q, K, V), ans, V, "device level")


@pytest.mark.parametrize("case", [c for c in golden_cases() if not c.get("file")], ids=lambda c: c["name"])
def test_mid_size_golden_runs_multi_split_launches_against_reference_bytes(case, pkg, be, O):
    """the small fixtures fit one tile or two; these (m = 256, n = 8192, d = 128 / 512, D3) make
    the launch cut K/V into several splits (or stream-K pieces) over 256 tiles, and the answer they are checked
    against was produced by the reference's own attention() (oracle/make_golden.py), not by our restatement."""
    Q, K, V, ans = O.load_golden(case)
    m, n, dk, dv = case["m"], case["n"], case["dk"], case["dv"]
    assert pkg.load().sdpa_dev_kv_splits(m, n, dk, dv) > 1
    e_dev = check(dev_attention(pkg, be, Q, K, V), ans, V, "device level")
    e_host = check(pkg.attention(Q, K, V), ans, V, "host level")
    idx = json.load(open(os.path.join(GOLD, "ref_mpi_fp32", "INDEX.json")))
    e_ref = max(e["max_abs_err_vs_fp64"] for e in idx if e["case"] == case["name"])
    print("%-14s err_gpu host %.2e dev %.2e | err_ref_mpi (max of P = 1, 8) %.2e" % (case["name"], e_host, e_dev, e_ref))
    assert max(e_dev, e_host) <= 4.0 * e_ref + 1e-7


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_golden_cli_prints_correct(case, O, tmp_path):
    """the plain-C host: same stdout contract as attention.c:184-189"""
    r = subprocess.run([CLI, O.golden_file(case, tmp_path)], capture_output=True, text=True,
                       env=dict(os.environ, SDPA_VERBOSE="1"))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")
    assert lines[0] == "Correct!" and lines[1].startswith("Elapsed time: ") and lines[1].endswith(" us")
    assert lines[2] == "" and len(lines) == 3
    assert "non-finite" not in r.stderr


CLI_MPI = os.path.join(ROOT, PKG, "bin", "attention-mpi-hip")
MPIEXEC = "/opt/conda/bin/mpiexec"


@pytest.mark.parametrize("ranks", [1, 4])
@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_golden_mpi_flavour_cli_prints_correct(case, ranks, O, tmp_path):
    """the MPI-flavour drop-in (attention-mpi.c:497-541): same stdout under mpiexec -n 1 and -n 4;
    only rank 0 touches the GPU"""
    if not (os.path.exists(CLI_MPI) and os.path.exists(MPIEXEC)):
        pytest.skip("no MPI in this image")
    r = subprocess.run([MPIEXEC, "-n", str(ranks), CLI_MPI, O.golden_file(case, tmp_path)],
                       capture_output=True, text=True, env=dict(os.environ, SDPA_VERBOSE="1"))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")
    assert lines[0] == "Correct!" and lines[1].startswith("Elapsed time: ") and lines[1].endswith(" us")
    assert lines[2] == "" and len(lines) == 3
    assert "%d MPI ranks" % ranks in r.stderr and "non-finite" not in r.stderr


def test_cli_virtual_ranks_and_plans(tmp_path, O):
    """the CLI's multi-rank switches on one device: K/V-sharded with both merges, and q-row-sharded"""
    case = os.path.join(GOLD, "adversarial_D4.bin")
    for env, tag in (({"SDPA_VIRTUAL_GPUS": "4"}, "merge=all-gather"),
                     ({"SDPA_VIRTUAL_GPUS": "4", "SDPA_MERGE": "allreduce"}, "merge=all-reduce x2"),
                     ({"SDPA_VIRTUAL_GPUS": "3", "SDPA_PLAN": "qrows"}, "plan=qrows")):
        r = subprocess.run([CLI, case], capture_output=True, text=True, env=dict(os.environ, SDPA_VERBOSE="1", **env))
        assert r.returncode == 0 and r.stdout.startswith("Correct!\nElapsed time: "), (env, r.stderr)
        assert tag in r.stderr and "(virtual)" in r.stderr, r.stderr


def test_cli_reports_wrong_on_a_corrupted_answer(tmp_path, O):
    Q, K, V, ans = O.read_case(os.path.join(GOLD, "tiny_D1.bin"))
    bad = ans.copy()
    bad[3, 2] += 0.5
    p = str(tmp_path / "bad.bin")
    O.write_case(p, Q, K, V, bad)
    r = subprocess.run([CLI, p], capture_output=True, text=True)
    assert r.returncode == 0
    assert r.stdout.startswith("Expect result[3][2] to be ") and r.stdout.endswith("Wrong!\n")


# ---------------------------------------------------------------- shape / edge matrix --------
SHAPES = [
    # m,    n,   dk,  dv, dist
    (1,     1,    1,   1, "D2"),      # smallest possible
    (1,   100,    4,   4, "D2"),
    (5,     3,    2,   7, "D1"),      # odd everything, n < tile
    (32,   32,   32,  32, "D2"),      # exactly one wave / one tile
    (33,   33,   33,  33, "D2"),      # one past every tile edge
    (127, 255,   64,  64, "D1"),
    (129, 257,   65,  31, "D3"),      # pads dk to 128, dv to 32
    (512, 512,   64,  64, "D1"),      # BASELINE config 1
    (300, 1000,  72,  40, "D2"),      # dims not multiples of 16/32
    (64,    5,   16,  16, "D2"),
    (257, 2048, 128, 128, "D3"),      # peaky softmax
    (256, 4096, 128, 128, "D4"),      # late spike key: running max jumps in the last tiles
    (130,  700,  96, 128, "D2"),
    (200,  333, 128,  64, "D1"),
    (40,   300, 100, 200, "D2"),      # dv > 128 -> any-shape kernel
    (24,   200, 256, 256, "D1"),      # dk, dv > 128
    (16,   130, 512, 512, "D1"),      # BASELINE config 5 dims in fp32: dk-split kernel, 128-col slices
    (70,   333, 300,  40, "D2"),      # 256 < dk <= 512: dk-split kernel, 32-col slices, ragged everything
    (129,  700, 512, 300, "D1"),      # dk-split, three q blocks of 64 (last ragged), waves past dv idle
    (64,  2048, 384, 128, "D3"),      # dk-split, peaky, one wave's dk slice all padding
    (200,    5, 512,  64, "D2"),      # dk-split, n < tile
    (33,    64, 257, 257, "D1"),      # one past both MFMA-kernel limits: 64-col slices
    (300, 4096, 512, 200, "D2"),      # dk-split with in-GPU K/V splits
    (130, 1000, 400, 700, "D4"),      # dk-split, dv > 512 -> two chunks, late spike key
    (20,    70, 600,  48, "D2"),      # dk > 512: VALU any-shape kernel
    (1,      1, 400,   1, "D2"),      # dk-split, smallest possible
    (65,    33, 260, 130, "D3"),      # dk-split, one row past a 64-row workgroup, one key past a tile
    (70,   333, 1024, 64, "D2"),      # 512 < dk <= 1024: dk-split kernel, 256-wide slices, one q block per workgroup
    (129,  700, 600, 300, "D1"),      # same, ragged dk (the last wave's slice mostly padding), 64-col dv slices
    (33,  2055, 1000, 1000, "D3"),    # same, dv > 512 -> two chunks, peaky, in-GPU splits
    (20,    70, 1100, 48, "D2"),      # dk > 1024: VALU any-shape kernel
    (40,   300, 100, 1500, "D2"),     # dv > 1024: twelve 128-column chunks (any dv while dk <= 1024)
    (70,   200, 600, 2100, "D1"),     # same on the one-block dk-split kernel: five 512-column chunks
    (300, 1000, 256, 256, "D2"),      # dense 256-wide: pipelined kernel, one wave per SIMD, ragged last tile
    (129, 2055, 256, 128, "D3"),      # same kernel family, dk = 256 / dv = 128, peaky
    (257,  700, 128, 256, "D4"),      # dk = 128 / dv = 256, late spike key
    (513, 8192, 256, 256, "D1"),      # several q blocks, in-GPU K/V splits
]


@pytest.mark.parametrize("m,n,dk,dv,dist", SHAPES)
def test_shapes_device_level(m, n, dk, dv, dist, pkg, be, orc, O):
    Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=m + n)
    check(dev_attention(pkg, be, Q, K, V), orc.attention_f64(Q, K, V), V, "device level")


@pytest.mark.parametrize("m,n,dk,dv,dist", SHAPES[3:14:2])
def test_shapes_host_level(m, n, dk, dv, dist, pkg, orc, O):
    Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=m + n)
    check(pkg.attention(Q, K, V), orc.attention_f64(Q, K, V), V, "host level")


def test_dv_beyond_1024_host_level(pkg, be, orc, O):
    """the reference takes any dv and any dk (dot_avx512 / axpy_avx512 loop over any n, attention-mpi.c:103-140); here the
    MFMA kernels chunk the value columns, the VALU any-shape kernel (dk > 1024) is launched once per 1024 columns and,
    beyond dk = 4096, reads its query row from global memory instead of LDS: nothing is refused (round 5)"""
    Q, K, V = O.make_inputs(50, 3000, 128, 1300, "D2", seed=8)
    check(pkg.attention(Q, K, V), orc.attention_f64(Q, K, V), V, "dv = 1300, streamed")
    Q, K, V = O.make_inputs(8, 40, 1100, 2500, "D1", seed=9)
    check(pkg.attention(Q, K, V), orc.attention_f64(Q, K, V), V, "dk = 1100, dv = 2500: three launches of the any-shape kernel")
    check(dev_attention(pkg, be, Q, K, V), orc.attention_f64(Q, K, V), V, "same, device level")
    Q, K, V = O.make_inputs(2, 3, 4100, 8, "D1", seed=10)
    check(pkg.attention(Q, K, V), orc.attention_f64(Q, K, V), V, "dk = 4100: Q rows from global memory")
    check(dev_attention(pkg, be, Q, K, V), orc.attention_f64(Q, K, V), V, "same, device level")
    Q, K, V = O.make_inputs(37, 700, 9000, 70, "D2", seed=11)
    check(pkg.attention(Q, K, V), orc.attention_f64(Q, K, V), V, "dk = 9000")


def test_bf16_beyond_its_kernels_dims_runs_the_fp32_path(pkg, orc, O, capfd):
    """SDPA_F_BF16 with dk > 512 or dv > 1024: the call answers on the fp32 path -- at the fp32 tolerance -- and says so on
    stderr, instead of SDPA_EUNSUP (VERDICT r4 "what's missing" 4: a drop-in should still answer)"""
    for (m, n, dk, dv) in [(8, 40, 600, 64), (20, 300, 64, 1100)]:
        Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=dk + dv)
        got = pkg.attention(Q, K, V, precision="bf16")
        check(got, orc.attention_f64(Q, K, V), V, "bf16 asked, fp32 run, dk=%d dv=%d" % (dk, dv))
        assert "runs the fp32 path" in capfd.readouterr().err
    Q, K, V = O.make_inputs(20, 300, 512, 1024, "D1", seed=3)                 # the largest dims the bf16 kernels take: bf16 it is
    got = pkg.attention(Q, K, V, precision="bf16")
    assert "runs the fp32 path" not in capfd.readouterr().err
    check(got, orc.attention_f64(Q, K, V), V, "bf16 at its limit", 1e-2 * max(1.0, float(np.abs(V).max())))


def test_host_level_q_pipeline_batches(pkg, orc, O, monkeypatch):
    """several ping-ponged Q batches with a ragged last one (attention-mpi.c:307-330)"""
    Q, K, V = O.make_inputs(1000, 600, 64, 64, "D2", seed=5)
    want = orc.attention_f64(Q, K, V)
    monkeypatch.setenv("SDPA_QBATCH", "192")
    got = pkg.attention(Q, K, V)
    assert pkg.last_timing()["q_batches"] == 6
    check(got, want, V, "6 batches")
    monkeypatch.delenv("SDPA_QBATCH")
    got1 = pkg.attention(Q, K, V, flags=1)      # SDPA_F_NO_PIPELINE
    assert pkg.last_timing()["q_batches"] == 1
    check(got1, want, V, "1 batch")


# ---------------------------------------------------------------- the shard-local triple -----
@pytest.mark.parametrize("m,n_local,dk,dv", [(100, 333, 64, 64), (70, 64, 128, 128), (513, 2000, 128, 128)])
def test_partial_triple_matches_oracle(m, n_local, dk, dv, pkg, be, orc, O):
    """contrib (un-normalised), lmax, lsum of attention-mpi.c:168-189, incl. in-GPU K/V splits"""
    Q, K, V = O.make_inputs(m, n_local, dk, dv, "D2", seed=3)
    Qf, Kf, Vf = (x.astype(np.float32) for x in (Q, K, V))
    c_ref, mx_ref, s_ref = orc.shard_partial_f32(Qf, Kf, Vf)
    sa = pkg.ShardedAttention(be)
    sa.load_kv_from_root(K, V, n_local, dk, dv)
    contrib, lmax, lsum = sa.batch_partial(be.cvt_d2f(torch.from_numpy(Q).cuda()))
    contrib, lmax, lsum = contrib.cpu().numpy()[:, :dv], lmax.cpu().numpy(), lsum.cpu().numpy()
    assert np.abs(lmax - mx_ref).max() <= 2e-6 * max(1.0, np.abs(mx_ref).max())
    assert np.abs(lsum / s_ref - 1).max() <= 2e-5
    assert np.abs(contrib - c_ref).max() <= 2e-5 * max(1.0, np.abs(c_ref).max())


def test_empty_shard_triple(pkg, be):
    """n_local = 0: contrib = 0, lmax = -inf, lsum = 0 (attention-mpi.c:172-173)"""
    qf = torch.randn(40, 64, device="cuda")
    kf = torch.empty(0, 64, device="cuda")
    vf = torch.empty(0, 64, device="cuda")
    contrib, lmax, lsum = be.shard_partial(qf, kf, vf, 64, 64)
    assert torch.all(contrib == 0) and torch.all(lsum == 0) and torch.all(torch.isneginf(lmax))


def test_kv_splits_are_used_and_agree(pkg, be, orc, O):
    """few query blocks + long K/V: the fused kernel splits K/V inside the GPU and merges"""
    m, n, dk, dv = 256, 8192, 128, 128
    assert pkg.load().sdpa_dev_kv_splits(m, n, dk, dv) > 1
    Q, K, V = O.make_inputs(m, n, dk, dv, "D3", seed=8)
    check(dev_attention(pkg, be, Q, K, V), O.numpy_attention_f64(Q, K, V), V, "kv splits")


# ---------------------------------------------------------------- shard merge (multi-GPU algebra)
@pytest.mark.parametrize("parts", [2, 3, 8])
def test_shard_merge_equals_single_shard(parts, pkg, be, orc, O):
    """P shards computed one after the other on one GPU and merged with the merge kernels
    (all-reduce MAX / SUM and reduce SUM done with torch ops) == the unsharded result.
    parts=8 with n=5 leaves three shards empty."""
    for (m, n, dk, dv, dist) in [(96, 1000, 64, 64, "D4"), (64, 5, 16, 16, "D2")]:
        Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=parts)
        want = orc.attention_f64(Q, K, V)
        qf = be.cvt_d2f(torch.from_numpy(Q).cuda())
        triples = []
        for r in range(parts):
            c, d = pkg.owner_count(n, parts, r), pkg.owner_disp(n, parts, r)
            sa = pkg.ShardedAttention(be)
            sa.load_kv_from_root(K[d:d + c], V[d:d + c], c, dk, dv)
            triples.append(sa.batch_partial(qf))
        gmax = torch.stack([t[1] for t in triples]).max(dim=0).values
        for contrib, lmax, lsum in triples:
            be.merge_rescale(contrib, lsum, lmax, gmax, dv)
        gsum = torch.stack([t[2] for t in triples]).sum(dim=0)
        for contrib, _, _ in triples:
            be.merge_normalise(contrib, gsum, dv)
        total = torch.stack([t[0] for t in triples]).sum(dim=0)
        got = be.cvt_f2d(total, dv).cpu().numpy()
        check(got, want, V, "P=%d" % parts)


# ---------------------------------------------------------------- converts -------------------
def test_cvt_d2f_is_round_to_nearest_even_and_pads_zero(be):
    x = torch.randn(37, 70, dtype=torch.float64, device="cuda") * 1e3
    x[0, 0] = 1.0 + 2.0 ** -24            # exactly halfway between two floats -> even
    x[0, 1] = 1.0 + 2.0 ** -24 + 2.0 ** -40
    y = be.cvt_d2f(x)
    assert y.shape == (37, 128)           # head dims in (32, 256] are padded to 64 / 128 / 256 zero-filled columns
    assert torch.equal(y[:, :70], x.to(torch.float32)) and torch.all(y[:, 70:] == 0)
    assert be.cvt_d2f(x[:, :30].contiguous()).shape == (37, 32) and be.cvt_d2f(torch.zeros(3, 300, dtype=torch.float64, device="cuda")).shape == (3, 300)
    assert y[0, 0].item() == 1.0 and y[0, 1].item() > 1.0
    z = be.cvt_f2d(y, 70)
    assert torch.equal(z, y[:, :70].to(torch.float64))
    big = torch.randn(1000, 128, dtype=torch.float64, device="cuda")
    assert torch.equal(be.cvt_d2f(big), big.to(torch.float32))


# ---------------------------------------------------------------- full-size properties -------
def test_config2_rows_and_properties(pkg, be, O):
    """BASELINE config 2 (m=n=8192, d=128): a row subset against the fp64 oracle, plus
    size-independent properties: linearity in V, invariance under a permutation of the K/V rows,
    rows of softmax weights summing to one (V = ones -> result = ones)."""
    m = n = 8192
    d = 128
    Q, K, V = O.make_inputs(m, n, d, d, "D2", seed=42)
    got = dev_attention(pkg, be, Q, K, V)
    rows = np.random.default_rng(0).choice(m, 192, replace=False)
    check(got[rows], O.numpy_attention_f64(Q, K, V, rows), V, "row subset")
    V2 = np.random.default_rng(1).standard_normal(V.shape)
    lin = dev_attention(pkg, be, Q, K, 2.0 * V - 0.5 * V2)
    got2 = dev_attention(pkg, be, Q, K, V2)
    assert np.abs(lin - (2.0 * got - 0.5 * got2)).max() <= 4 * fp32_tol(V)
    perm = np.random.default_rng(2).permutation(n)
    assert np.abs(dev_attention(pkg, be, Q, K[perm], V[perm]) - got).max() <= 2 * fp32_tol(V)
    ones = dev_attention(pkg, be, Q, K, np.ones_like(V))
    assert np.abs(ones - 1.0).max() <= 1e-5


def test_headline_shape_row_subset(pkg, O):
    """the metric shape m=32768 n=65536 d=128 through the host-level boundary; 128 random rows
    against the fp64 oracle and the idempotence of a second call"""
    m, n, d = 32768, 65536, 128
    Q, K, V = O.make_inputs(m, n, d, d, "D1", seed=7)
    got = pkg.attention(Q, K, V)
    assert np.isfinite(got).all()
    rows = np.random.default_rng(3).choice(m, 128, replace=False)
    check(got[rows], O.numpy_attention_f64(Q, K, V, rows), V, "headline rows")
    t = pkg.last_timing()
    assert t["n_gpus"] >= 1 and t["q_batches"] == 1      # default Q batch = 32768 rows
    # round 5: ONE streamed launch that follows 5 K/V groups (rounds 1-4: a launch per chunk, first and last chunk in 4 row
    # pieces -- kv_chunks + 6 launches; $SDPA_STREAMED=0 still runs that)
    assert t["streamed"] == 1 and t["kv_chunks"] >= 4 and t["fused_launches"] == 1, t
    assert t["last_kernel"] == "sdpa::fused_pipelined_stream_kernel<128,128>", t
    again = pkg.attention(Q, K, V)
    assert np.array_equal(again, got), "same inputs must give bit-identical results run to run"


def test_qrows_plan_single_rank(pkg, orc, O):
    """the Q-row-sharded alternative plan at world size 1 (engine.attention_qrows)"""
    Q, K, V = O.make_inputs(150, 400, 64, 64, "D2", seed=12)
    got = pkg.attention_qrows(Q, K, V, 150, 400, 64, 64, 0, 1)
    check(got, orc.attention_f64(Q, K, V), V, "qrows")


def test_cli_bf16_env_and_qbatch(tmp_path, O):
    """the CLI's environment switches: bf16 operands still pass the template's 0.02 check, and a
    small SDPA_QBATCH runs the multi-batch pipeline"""
    Q, K, V, ans = O.read_case(os.path.join(GOLD, "cfg1_small_D1.bin"))
    for env in ({"SDPA_PRECISION": "bf16"}, {"SDPA_QBATCH": "32"}):
        r = subprocess.run([CLI, os.path.join(GOLD, "cfg1_small_D1.bin")], capture_output=True, text=True,
                           env=dict(os.environ, SDPA_VERBOSE="1", **env))
        assert r.returncode == 0 and r.stdout.startswith("Correct!\nElapsed time: "), r.stderr
        if "SDPA_QBATCH" in env:
            assert "q_batches=3" in r.stderr


def test_cli_io_modes_and_pinned_host_arrays(tmp_path, pkg, orc, O):
    """the CLI reads into page-locked memory by default, into malloc'd memory with
    $SDPA_DEBUG=pinned_io=0, and creates the engine inside the timed call with $SDPA_DEBUG=time_init=1 -- same
    verdict each way; sdpa_host_alloc memory works as caller arrays of the boundary call"""
    case = os.path.join(GOLD, "cfg1_small_D1.bin")
    for env in ({}, {"SDPA_DEBUG": "pinned_io=0"}, {"SDPA_DEBUG": "time_init=1"}, {"SDPA_CLI_PREFETCH": "1"},
                {"SDPA_CLI_PREFETCH": "1", "SDPA_DEBUG": "pinned_io=0", "SDPA_VIRTUAL_GPUS": "2"}):
        r = subprocess.run([CLI, case], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout.startswith("Correct!\nElapsed time: "), (env, r.stderr)
    import ctypes
    lib = pkg.load()
    Q, K, V = O.make_inputs(64, 2048, 128, 128, "D2", seed=77)       # K, V: 2 MiB each
    bufs = []
    def pinned_copy(a):
        p = lib.sdpa_host_alloc(a.nbytes)
        assert p, "sdpa_host_alloc returned NULL on a GPU box"
        bufs.append(p)
        out = np.ctypeslib.as_array((ctypes.c_double * a.size).from_address(p)).reshape(a.shape)
        out[...] = a
        return out
    try:
        Kp, Vp = pinned_copy(K), pinned_copy(V)
        got = pkg.attention(Q, Kp, Vp)
        assert np.abs(got - orc.attention_f64(Q, K, V)).max() <= fp32_tol(V)
    finally:
        for p in bufs:
            lib.sdpa_host_free(p)
    lib.sdpa_host_free(None)
    assert lib.sdpa_host_alloc(0) is None


def test_random_shape_sweep_f32(pkg, be, orc, O):
    """40 seeded random shapes (ragged everything, dims on both sides of every kernel's tile and
    dispatch boundaries) through the device-level path"""
    rng = np.random.default_rng(2024)
    worst = 0.0
    for it in range(40):
        m = int(rng.integers(1, 300))
        n = int(rng.integers(1, 700))
        dk = int(rng.choice([1, 7, 16, 31, 32, 33, 64, 65, 96, 127, 128, 129, 160]))
        dv = int(rng.choice([1, 5, 16, 32, 33, 64, 72, 100, 128, 130, 192]))
        dist = ["D1", "D2", "D3", "D4"][it % 4]
        Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=1000 + it)
        got = dev_attention(pkg, be, Q, K, V)
        want = orc.attention_f64(Q, K, V)
        assert np.isfinite(got).all(), (m, n, dk, dv, dist)
        rel = np.abs(got - want).max() / fp32_tol(V)
        assert rel <= 1.0, "shape %s: err/tol = %.3f" % ((m, n, dk, dv, dist), rel)
        worst = max(worst, rel)
    print("worst err/tol over the sweep: %.3f" % worst)


def test_error_ratio_vs_reference_mpi_program_outputs(pkg, be, O):
    """err_gpu / err_ref_mpi with err_ref_mpi taken from RAW OUTPUTS of the
    reference's own MPI program (tests/golden/ref_mpi_fp32/, attention-mpi.c unmodified at
    P = 1, 2, 8): the HIP path's error against the fp64 answer must be of the same order."""
    idx = json.load(open(os.path.join(GOLD, "ref_mpi_fp32", "INDEX.json")))
    for case in golden_cases():
        Q, K, V, ans = O.load_golden(case)
        e_ref = []
        for e in idx:
            if e["case"] == case["name"]:
                ref = np.fromfile(os.path.join(GOLD, "ref_mpi_fp32", e["file"]), dtype=np.float32)
                e_ref.append(np.abs(ref.reshape(ans.shape).astype(np.float64) - ans).max())
        assert len(e_ref) == (3 if case.get("file") else 2)      # (the mid-size cases: P = 1 and 8)
        e_host = np.abs(pkg.attention(Q, K, V) - ans).max()
        e_dev = np.abs(dev_attention(pkg, be, Q, K, V) - ans).max()
        print("%-16s err_gpu host %.2e dev %.2e | err_ref_mpi P=1,2,8 %s | ratio %.2f" % (
            case["name"], e_host, e_dev, " ".join("%.2e" % x for x in e_ref), max(e_host, e_dev) / max(e_ref)))
        assert max(e_host, e_dev) <= 4.0 * max(e_ref) + 1e-7


def test_error_ratio_vs_reference_fp32_pipeline(pkg, be, orc, O):
    """the same ratio at larger shapes, err_ref from the restated fp32 pipeline -- which is pinned
    BIT FOR BIT to the reference's MPI program at P = 1, 2, 8 (tests/test_oracle.py); expected
    ratio <~ 2"""
    for (m, n, d, dist) in [(200, 4096, 128, "D3"), (300, 1000, 64, "D2"), (128, 2048, 128, "D1")]:
        Q, K, V = O.make_inputs(m, n, d, d, dist, seed=31)
        want = orc.attention_f64(Q, K, V)
        e_gpu = np.abs(dev_attention(pkg, be, Q, K, V) - want).max()
        e_ref = max(np.abs(orc.attention_sharded_f32(Q, K, V, p) - want).max() for p in (1, 8))
        print("m=%d n=%d d=%d %s: err_gpu %.2e  err_ref_fp32 %.2e  ratio %.2f" % (m, n, d, dist, e_gpu, e_ref, e_gpu / e_ref))
        assert e_gpu <= 4.0 * e_ref + 1e-7


def test_race_screen_repeatability(pkg, be, orc, O):
    """The pipelined kernel hands K/V tiles between waves through LDS-DMA + barriers, and the partial
    triples of its in-GPU K/V splits between WORKGROUPS (different XCDs, non-coherent L2s) through an
    agent-scope release / ticket / acquire when the kernel merges them itself ($SDPA_DEBUG=split_merge=kernel,
    the second half of this screen); a missing wait or a stale line shows up as rare wrong tiles.
    Screen: 100 launches each of three shapes (in-GPU splits, ragged tails, both MFMA kernels) must be
    BITWISE identical to the first, which is checked against the oracle.  All shapes share one scratch
    area, so every launch finds the previous shape's arrival words in it (another generation: they
    must count as zero)."""
    shapes = [(640, 6000, 128, 128, "D2"), (513, 3333, 64, 64, "D3"), (300, 2500, 96, 72, "D2")]
    runs = []
    for (m, n, dk, dv, dist) in shapes:
        assert pkg.load().sdpa_dev_kv_splits(m, n, dk, dv) > 1
        Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=77)
        sa = pkg.ShardedAttention(be)
        sa.load_kv_from_root(K, V, n, dk, dv)
        qf = sa.convert_q(torch.from_numpy(Q).cuda())
        contrib, lmax, lsum = sa.batch_partial(qf)
        first = (contrib.clone(), lmax.clone(), lsum.clone())
        check(be.finish_f64(contrib, lsum, dv).cpu().numpy(), orc.attention_f64(Q, K, V), V, "first launch")
        runs.append((sa, qf, dv, first))
    for it in range(1, 100):
        if it == 50:
            os.environ["SDPA_DEBUG"] = "split_merge=kernel"
            pkg.reload_env()
        for sa, qf, dv, first in runs:                 # interleaved: the scratch area changes hands every launch
            cur = sa.batch_partial(qf)
            assert all(torch.equal(a_[:, :dv] if a_.dim() == 2 else a_, b_[:, :dv] if b_.dim() == 2 else b_)
                       for a_, b_ in zip(cur, first)), "launch %d differs from launch 0" % it
    os.environ.pop("SDPA_DEBUG", None)
    pkg.reload_env()


@pytest.mark.parametrize("m,n,dk,dv", [(8192, 8192, 128, 128),      # BASELINE config 2: 64 query blocks x 8 splits
                                        (640, 6000, 128, 128), (513, 3333, 64, 64), (300, 2500, 96, 72),
                                        (1000, 20000, 256, 256),     # one wave per SIMD variant
                                        (4096, 8192, 128, 64), (700, 9000, 64, 128), (900, 7000, 250, 120)])
def test_in_kernel_split_merge_equals_the_separate_pass_bitwise(m, n, dk, dv, pkg, be, O, monkeypatch):
    """kv_splits > 1 on the pipelined kernels: split_merge_kernel merges the partial triples right behind
    the fused launch (the default: measured faster).  With $SDPA_DEBUG=split_merge=kernel the last workgroup
    of a query block to arrive merges the block's triples inside the fused launch (one launch per step):
    same weights, same sums in the same split order -- the two forms must agree bit for bit, 20 launches
    (under load from the neighbouring query blocks)."""
    assert pkg.load().sdpa_dev_kv_splits(m, n, dk, dv) > 1
    Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=5)
    sa = pkg.ShardedAttention(be)
    sa.load_kv_from_root(K, V, n, dk, dv)
    qf = sa.convert_q(torch.from_numpy(Q).cuda())
    monkeypatch.delenv("SDPA_DEBUG", raising=False)
    pkg.reload_env()
    want = tuple(t.clone() for t in sa.batch_partial(qf))
    monkeypatch.setenv("SDPA_DEBUG", "split_merge=kernel")
    pkg.reload_env()
    for it in range(20):
        got = sa.batch_partial(qf)
        for name, g, w in zip(("contrib", "lmax", "lsum"), got, want):
            g, w = (g[:, :dv], w[:, :dv]) if g.dim() == 2 else (g, w)
            assert torch.equal(g, w), "launch %d: %s of the in-kernel merge differs from the separate pass" % (it, name)
    assert np.isfinite(be.finish_f64(got[0], got[2], dv).cpu().numpy()).all()


@pytest.mark.parametrize("m,n,dk,dv,dist", [
    (260, 5000, 512, 512, "D2"),       # BASELINE config 5's dims: ragged rows, in-GPU splits, ragged last tile
    (64, 32, 512, 512, "D1"),          # one tile
    (65, 33, 512, 512, "D2"),          # two tiles, the second one key long
    (130, 64, 384, 384, "D3"),         # two full tiles; 96-wide dk slices
    (100, 95, 320, 264, "D4"),         # three tiles, the last ragged
    (257, 1000, 768, 768, "D2"),       # one query block per workgroup, two dv chunks
    (96, 777, 1024, 32, "D2"),         # 256-wide slices, ONE P.V MFMA per k-step: four units per gap
    (1024, 2049, 512, 200, "D3"),      # 64-wide dv slices (two units per gap), dv not a multiple of the chunk
    (96, 777, 700, 130, "D2"),
    (200, 3000, 200, 256, "D2"),       # 128 < dk <= 256 with dv = 256: NOT this kernel's (fused_pipelined_kernel<256,256>), same bar
    (200, 3000, 300, 256, "D2"),       # just beyond: 96-wide dk slices of a padded dk = 384
    (33, 1, 640, 640, "D1"),           # a single key
])
def test_f32_dksplit_pipelined_kernel_against_the_oracle_and_itself(m, n, dk, dv, dist, pkg, be, orc, O):
    """dk > 256 in fp32: fused_dksplit_pipe_kernel places the exchange sums and the softmax of tile t+1 between the P.V MFMAs of
    tile t (its serial-phase twin was retired in round 6).  Its partial score tiles cross LDS behind one barrier per tile: a
    missing wait shows as rare wrong tiles -- 5 launches must be IDENTICAL -- and the result sits within the fp32 tolerance of
    the fp64 oracle."""
    Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed=m + n + dk)
    sa = pkg.ShardedAttention(be)
    sa.load_kv_from_root(K, V, n, dk, dv)
    qf = sa.convert_q(torch.from_numpy(np.ascontiguousarray(Q)).cuda())
    want = tuple(t.clone() for t in sa.batch_partial(qf))
    assert ("fused_dksplit_pipe_kernel" if dk > 256 else "fused_pipelined_kernel<256,256") in pkg.last_launch()["kernel"]
    for it in range(5):
        got = sa.batch_partial(qf)
        for name, g, w in zip(("contrib", "lmax", "lsum"), got, want):
            g, w = (g[:, :dv], w[:, :dv]) if g.dim() == 2 else (g, w)
            assert torch.equal(g, w), "launch %d: %s differs from the first launch" % (it, name)
    res = be.finish_f64(got[0], got[2], dv).cpu().numpy()
    check(res, orc.attention_f64(Q, K, V), V, "dk-split pipelined")


@pytest.mark.parametrize("m,n,dk,dv", [(70, 200, 600, 2100),      # 256-wide dv slices want rows of 8 k floats: 2100 is not
                                        (90, 500, 320, 320),       # 96-wide slices want rows of 12 k: 320 is not
                                        (64, 300, 700, 700)])      # 192-wide slices want rows of 12 k: 700 is not
def test_f32_dksplit_kernels_take_any_leading_dimension_the_api_accepts(m, n, dk, dv, pkg, be, orc, O):
    """the device-level entry point accepts any image leading dimension that is a multiple of 4.  The dk-split
    kernels' matched dv slices read runs of 3 / 6 / 8 consecutive V columns per lane, which must not straddle a
    row end: dense_ld() images never do, for any other leading dimension the launch takes the 128-wide slices
    (runs of 4).  Images built by hand at round4(d) against the images cvt_d2f builds, and the fp64 oracle."""
    Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=dk + dv)
    want = orc.attention_f64(Q, K, V)

    def image(x, ld):
        t = torch.zeros((x.shape[0], ld), dtype=torch.float32, device="cuda")
        t[:, :x.shape[1]] = torch.from_numpy(x).cuda().to(torch.float32)
        return t
    r4 = lambda d: (d + 3) // 4 * 4
    assert r4(dv) != pkg.load().sdpa_dev_dense_ld(dv)
    for ldq, ldk, ldv in ((r4(dk), r4(dk), r4(dv)), (r4(dk) + 4, r4(dk) + 8, r4(dv) + 4)):
        contrib, lmax, lsum = be.shard_partial(image(Q, ldq), image(K, ldk), image(V, ldv), dk, dv)
        check(be.finish_f64(contrib, lsum, dv).cpu().numpy(), want, V, "hand-built images, ld %d / %d / %d" % (ldq, ldk, ldv))
    check(dev_attention(pkg, be, Q, K, V), want, V, "dense_ld images")


def steep_late_rise_inputs(m, n, d, vscale, seed=3, first=32, rise_nats=15.0):
    """Keys [0, first) score `rise_nats` BELOW all the others for every query, and every V entry is
    ~ +vscale: a kernel that defers its accumulator rescale (rise < 2^24) carries weights of e^15
    on n - first keys -- beyond fp32 for vscale * n * e^15 > 3.4e38 -- where the reference's eager
    rescale (attention-mpi.c:179-182) keeps every weight <= 1."""
    rng = np.random.default_rng(seed)
    u = rng.standard_normal(d)
    u /= np.linalg.norm(u)
    ab = rise_nats * np.sqrt(d) / 2.0
    a = b = np.sqrt(ab)
    Q = a * u[None, :] + 0.01 * rng.standard_normal((m, d))
    K = b * u[None, :] + 0.01 * rng.standard_normal((n, d))
    K[:first] = -b * u[None, :] + 0.01 * rng.standard_normal((first, d))
    V = vscale * (1.0 + 0.1 * rng.random((n, d)))
    return Q, K, V


@pytest.mark.parametrize("m,n,d,vscale", [(256, 65536, 128, 1e30),     # the verdict's case: |V| ~ 1e30, n = 65536, d = 128
                                           (2048, 16384, 128, 1e31),    # fewer splits per query block
                                           (256, 8192, 256, 1e32),      # one wave per SIMD variant
                                           (300, 8000, 64, 1e32)])
def test_fp32_range_where_the_deferred_rescale_would_overflow(m, n, d, vscale, pkg, be, orc, O):
    """attention-mpi.c:179-182 rescales on every new maximum, so its un-normalised sums stay below
    n * max|V|.  The pipelined kernels defer the rescale (weights up to 2^24): on these inputs their
    first pass overflows, the epilogue's range check sees it and the workgroup redoes its K/V range
    with an eager rescale.  Device level and boundary, against the fp64 oracle at the fp32 tolerance;
    the reference's own fp32 pipeline (oracle_attention_sharded_f32) is finite on the same inputs."""
    Q, K, V = steep_late_rise_inputs(m, n, d, vscale)
    want = O.numpy_attention_f64(Q, K, V)
    if m * n <= 256 * 65536:
        ref32 = orc.attention_sharded_f32(Q, K, V, 1)
        assert np.isfinite(ref32).all() and np.abs(ref32 - want).max() <= fp32_tol(V)
    # lazily rescaled weights alone would overflow: e^15 * (keys of split 0 beyond the first tile) * vscale
    sa = pkg.ShardedAttention(be)
    sa.load_kv_from_root(K, V, n, d, d)
    contrib, lmax, lsum = sa.batch_partial(sa.convert_q(torch.from_numpy(Q).cuda()))
    assert torch.isfinite(contrib).all() and torch.isfinite(lsum).all()
    check(be.finish_f64(contrib, lsum, d).cpu().numpy(), want, V, "device level")
    check(pkg.attention(Q, K, V), want, V, "boundary")
    # and as 4 loopback-free K/V shards merged with the reference's algebra (the shard holding the low keys redoes)
    parts = 4
    triples = []
    qf = be.cvt_d2f(torch.from_numpy(Q).cuda())
    for r in range(parts):
        c, dsp = pkg.owner_count(n, parts, r), pkg.owner_disp(n, parts, r)
        sh = pkg.ShardedAttention(be)
        sh.load_kv_from_root(K[dsp:dsp + c], V[dsp:dsp + c], c, d, d)
        triples.append(sh.batch_partial(qf))
    stats = torch.stack([torch.stack((t[1], t[2])) for t in triples]).contiguous()
    for r, (c_, _, _) in enumerate(triples):
        be.merge_gathered(c_, stats, r, d)
    check(be.cvt_f2d(torch.stack([t[0] for t in triples]).sum(dim=0), d).cpu().numpy(), want, V, "4 shards merged")


def test_gathered_merge_equals_two_phase(pkg, be, orc, O):
    """sdpa_dev_merge_gathered (one pass from all-gathered (lmax,lsum)) == rescale + normalise"""
    m, n, dk, dv, parts = 96, 1000, 64, 64, 5
    Q, K, V = O.make_inputs(m, n, dk, dv, "D4", seed=9)
    want = orc.attention_f64(Q, K, V)
    qf = be.cvt_d2f(torch.from_numpy(Q).cuda())
    triples = []
    for r in range(parts):
        c, d = pkg.owner_count(n, parts, r), pkg.owner_disp(n, parts, r)
        sa = pkg.ShardedAttention(be)
        sa.load_kv_from_root(K[d:d + c], V[d:d + c], c, dk, dv)
        triples.append(sa.batch_partial(qf))
    triples.append(be.shard_partial(qf, torch.empty(0, dk, device="cuda"), torch.empty(0, dv, device="cuda"), dk, dv))
    stats = torch.stack([torch.stack((t[1], t[2])) for t in triples]).contiguous()    # [P+1, 2, m]
    for r, (contrib, _, _) in enumerate(triples):
        be.merge_gathered(contrib, stats, r, dv)
    got = be.cvt_f2d(torch.stack([t[0] for t in triples]).sum(dim=0), dv).cpu().numpy()
    check(got, want, V, "gathered merge, one empty shard")


@pytest.mark.parametrize("m,n,dk,dv", [(192, 4096, 64, 64), (160, 4096, 128, 128), (100, 3000, 48, 40), (64, 2048, 200, 136),
                                       (160, 4096, 256, 256), (130, 2048, 256, 128), (130, 2048, 128, 256)])
def test_steep_softmax_exercises_the_rescale_paths(m, n, dk, dv, pkg, be, orc, O):
    """Scores with a standard deviation of ~25: the row max keeps jumping by far more than the
    deferred-rescale threshold (2^24) between tiles, so the rare rescale branches of every kernel
    (pipelined, register-staged, chunked) run many times; the softmax is nearly an arg-max."""
    rng = np.random.default_rng(dk + dv)
    s = 5.0 * (dk / 16.0) ** 0.25            # q.k/sqrt(dk) has std s^2 * ... ~ 25 at any dk
    Q = rng.standard_normal((m, dk)) * s
    K = rng.standard_normal((n, dk)) * s
    V = rng.standard_normal((n, dv)) * 3.0
    scores = (Q @ K.T) / np.sqrt(dk)
    assert scores.max(axis=1).min() - scores[:, :32].max(axis=1).max() < 0 or True   # (documentation only)
    assert (scores.max(axis=1) - scores[:, :32].max(axis=1)).max() > 17.0           # > 2^24 in the exp2 domain
    got = dev_attention(pkg, be, Q, K, V)
    check(got, orc.attention_f64(Q, K, V), V, "steep")
    # and through the host-level path with several in-GPU splits / batches
    check(pkg.attention(Q, K, V), orc.attention_f64(Q, K, V), V, "steep host")


@pytest.mark.parametrize("m,n,dk,dv", [(8192, 8192, 128, 128), (700, 9000, 72, 40), (300, 5000, 200, 300), (4096, 4096, 64, 128), (256, 64, 128, 128)])
def test_single_shard_call_with_the_fused_finish_equals_the_separate_kernels(m, n, dk, dv, pkg, be, O):
    """sdpa_dev_shard_attention_f64 (round 6): the fused kernel, then ONE pass that merges the in-GPU splits, normalises (merge step 5
    with gsum = lsum, attention-mpi.c:358-362) and writes fp64 (:373) -- against sdpa_dev_shard_partial_f32 + sdpa_dev_finish_f64
    (split_merge, then finish): the same rows bit for bit, with and without in-GPU splits, padded dims included."""
    Q, K, V = O.make_inputs(m, n, dk, dv, "D2", seed=m + n + dk)
    sa = pkg.ShardedAttention(be)
    sa.load_kv_from_root(K, V, n, dk, dv)
    qf = sa.convert_q(torch.from_numpy(Q).cuda())
    contrib, lmax, lsum = sa.batch_partial(qf)
    want = be.finish_f64(contrib, lsum, dv)
    got = sa.batch_attention_f64(qf)
    assert torch.equal(got, want), "fused finish differs in %d values" % int((got != want).sum())
    check(got.cpu().numpy(), O.numpy_attention_f64(Q, K, V), V, "single-shard call")


def test_batched_convert_writes_the_single_converters_images(be):
    """sdpa_dev_cvt_d2f_batch: three matrices of different shapes (dense and padded rows) in one launch == three sdpa_dev_cvt_d2f"""
    xs = [torch.randn(r, c, dtype=torch.float64, device="cuda") * 3 for r, c in ((8192, 128), (700, 72), (33, 300))]
    for got, x in zip(be.cvt_d2f_batch(xs), xs):
        assert torch.equal(got, be.cvt_d2f(x))
    assert torch.equal(be.cvt_d2f_batch(xs[:1])[0], be.cvt_d2f(xs[0]))
