"""CPU: the C-ABI library loads, exports every symbol include/sdpa_hip.h declares, validates its
arguments, and fails LOUDLY (no fallback) where there is no GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, PKG

HAS_GPU = torch.cuda.is_available()


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load()
    names = pkg.header_symbols()
    assert len(names) >= 17 and "sdpa_attention_f64" in names and "sdpa_dev_shard_partial_f32" in names
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    # and nothing is bound in Python that the header does not declare
    assert set(pkg._lib._PROTOS) == set(names)


def test_version_and_strerror(pkg):
    lib = pkg.load()
    assert b"gfx950" in lib.sdpa_version()
    assert lib.sdpa_strerror(0) == b"ok"
    assert b"no CPU fallback" in lib.sdpa_strerror(pkg._lib.SDPA_ENODEV)


@pytest.mark.parametrize("n,size", [(10, 3), (5, 8), (262144, 8), (100, 64)])
def test_owner_partition_matches_oracle(n, size, pkg, orc):
    for r in range(size):
        assert pkg.owner_count(n, size, r) == orc.owner_count(n, size, r)
        assert pkg.owner_disp(n, size, r) == orc.owner_disp(n, size, r)


def test_dense_ld_python_and_c_agree(pkg):
    """leading dimension of the fp32 operand images: (32, 256] padded to 64 / 128 / 256, up to 32 rounded to 4,
    beyond 256 to 12 / 4 / 12 / 8 (dv <= 384 / 512 / 768 / more: a lane of the dk-split kernels reads 3 / 4 / 6 / 8
    consecutive V columns, and no run may straddle the row end)"""
    lib = pkg.load()
    from importlib import import_module
    eng = import_module(pkg.__name__ + ".engine")
    for d in range(1, 700):
        assert lib.sdpa_dev_dense_ld(d) == eng.dense_ld(d) >= d and eng.dense_ld(d) % 4 == 0
    assert [lib.sdpa_dev_dense_ld(d) for d in (1, 32, 33, 64, 65, 100, 128, 129, 256, 257)] == \
        [4, 32, 64, 64, 128, 128, 128, 256, 256, 264]
    assert [lib.sdpa_dev_dense_ld(d) for d in (320, 384, 385, 512, 513, 640, 768, 769, 1024, 2100)] == \
        [324, 384, 388, 512, 516, 648, 768, 776, 1024, 2104]
    assert lib.sdpa_dev_dense_ld(0) == 0


def test_bf16_image_geometry_helpers(pkg):
    """host-side layout contract of the bf16 operand images (include/sdpa_hip.h): padded leading
    dimensions, the key order of a Vt row, the workspace of the wide (dv > 256) kernel"""
    lib = pkg.load()
    assert [lib.sdpa_dev_bf16_ld(d) for d in (1, 64, 65, 128, 200, 256, 300, 512)] == [64, 64, 128, 128, 256, 256, 512, 512]
    assert [lib.sdpa_dev_bf16_dvp(d) for d in (1, 64, 100, 128, 200, 256, 257, 512, 700, 1024)] == \
        [64, 64, 128, 128, 256, 256, 512, 512, 1024, 1024]
    assert [lib.sdpa_dev_bf16_ldn(n) for n in (0, 1, 32, 33, 65536)] == [0, 32, 32, 64, 65536]
    pos = [lib.sdpa_dev_bf16_kvpos(j) for j in range(64)]
    assert pos[:16] == [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]       # group order 0-3, 8-11, 4-7, 12-15
    assert all(pos[16 * g + i] == 16 * g + pos[i] for g in range(4) for i in range(16))
    assert [pos[p] for p in pos] == list(range(64))                                   # an involution
    # the eight keys one MFMA lane multiplies -- rows crow(8h+j, hi) of the score tile's C layout --
    # are positions 16h + 8hi .. +7 of the row, i.e. 16 contiguous bytes
    crow = lambda r, hi: (r & 3) + 8 * (r >> 2) + 4 * hi
    for h in range(2):
        for hi in range(2):
            assert [pos[crow(8 * h + j, hi)] for j in range(8)] == list(range(16 * h + 8 * hi, 16 * h + 8 * hi + 8))
    assert lib.sdpa_dev_bf16_kvpos(-1) < 0
    # split buffers ([splits x m x (ld(dv) + 2)] floats) when the shard is split in-GPU, and one
    # redo flag per (split, 128-row q block) for the kernels with a fixed reference exponent
    # (dv > 256: wide kernel; dk, dv <= 256: duo kernel)
    s1 = lib.sdpa_dev_kv_splits_bf16(32768, 65536, 128, 128)
    assert lib.sdpa_dev_workspace_bytes_bf16(32768, 65536, 128, 128) == (s1 * 32768 * (128 + 2) * 4 if s1 > 1 else 0) + 256 * s1 * 4
    s0 = lib.sdpa_dev_kv_splits_bf16(32768, 65536, 512, 256)      # general kernel: no flags
    assert lib.sdpa_dev_workspace_bytes_bf16(32768, 65536, 512, 256) == (s0 * 32768 * (256 + 2) * 4 if s0 > 1 else 0)
    # 274 workgroups of 256 rows on 256 CUs would run 1.07 rounds: two splits fill the last round better
    s2 = lib.sdpa_dev_kv_splits_bf16(70000, 4096, 64, 64)
    assert s2 == 2 and lib.sdpa_dev_workspace_bytes_bf16(70000, 4096, 64, 64) == s2 * 70000 * (64 + 2) * 4 + 547 * s2 * 4
    assert lib.sdpa_dev_kv_splits_bf16(65536, 4096, 64, 64) == 1 and lib.sdpa_dev_workspace_bytes_bf16(65536, 4096, 64, 64) == 512 * 4
    # fp32: one round of workgroups keeps the count that fills the chip once; beyond one round the last round is filled
    # (classic equal splits, $SDPA_DEBUG=streamk=0; the default since round 4 is stream-K for such shapes -- next test)
    os.environ["SDPA_DEBUG"] = "streamk=0"
    pkg.reload_env()
    try:
        assert lib.sdpa_dev_kv_splits(40000, 65536, 128, 128) == 8       # 313 query blocks: 2 splits = 1.22 rounds (61 %), 8 = 4.9 (98 %)
        assert lib.sdpa_dev_kv_splits(32768, 65536, 256, 256) == 1 and lib.sdpa_dev_kv_splits(40000, 65536, 256, 256) == 4
    finally:
        os.environ.pop("SDPA_DEBUG", None)
        pkg.reload_env()
    splits = lib.sdpa_dev_kv_splits_bf16(32768, 65536, 512, 512)
    assert splits == 1 and lib.sdpa_dev_workspace_bytes_bf16(32768, 65536, 512, 512) == 256 * 4
    s2 = lib.sdpa_dev_kv_splits_bf16(256, 8192, 512, 512)
    assert s2 > 1
    assert lib.sdpa_dev_workspace_bytes_bf16(256, 8192, 512, 512) == s2 * 256 * (512 + 2) * 4 + 2 * s2 * 4


def test_stream_k_plan_arithmetic(pkg):
    """Round 4: the fp32 pipelined kernels cut a launch's (query block, K/V tile) steps into equal runs, one per
    resident workgroup slot, when the classic grid of equal splits would not fill whole rounds.  Pure host
    arithmetic (sdpa_dev_kv_splits = slabs on a whole chip, sdpa_dev_workspace_bytes = scratch for ANY stream)."""
    lib = pkg.load()
    ws = lambda m, d, s: s * m * (d + 2) * 4 + (m + 127) // 128 * 8 if s > 1 else 0
    # shapes whose classic grid is exactly one round keep it: 256 query blocks x 2, 64 x 8, 1024 x 1 (two rounds)
    assert lib.sdpa_dev_kv_splits(32768, 65536, 128, 128) == 2
    assert lib.sdpa_dev_kv_splits(8192, 8192, 128, 128) == 8
    assert lib.sdpa_dev_kv_splits(131072, 65536, 128, 128) == 1
    # ... but their scratch covers the launch on a CU-masked stream too, where the same shape is cut by stream-K
    # into at most 3 pieces per query block (496 runs of 1058 tile steps over rows of 2048)
    assert lib.sdpa_dev_workspace_bytes(32768, 65536, 128, 128) == ws(32768, 128, 3)
    assert ws(131072, 128, 2) <= lib.sdpa_dev_workspace_bytes(131072, 65536, 128, 128) <= ws(131072, 128, 3)
    # 258 query blocks (m = 33000): classic needs many splits to fill rounds, stream-K cuts 258 x 2048 steps into 512 runs of 1032
    assert lib.sdpa_dev_kv_splits(33000, 65536, 128, 128) == 3
    assert lib.sdpa_dev_kv_splits(40000, 65536, 128, 128) == 3
    # head dims outside the pipelined kernels are untouched (dk-split kernel, bf16 has its own picker)
    os.environ["SDPA_DEBUG"] = "streamk=0"
    pkg.reload_env()
    try:
        classic = [lib.sdpa_dev_kv_splits(m, 65536, 512, 512) for m in (32768, 33000, 40000)]
        assert lib.sdpa_dev_kv_splits(33000, 65536, 128, 128) > 3
        assert lib.sdpa_dev_workspace_bytes(32768, 65536, 128, 128) >= ws(32768, 128, 2)    # (masked streams: many equal splits)
    finally:
        os.environ.pop("SDPA_DEBUG", None)
        pkg.reload_env()
    assert [lib.sdpa_dev_kv_splits(m, 65536, 512, 512) for m in (32768, 33000, 40000)] == classic


def test_declared_pinned_ranges_need_no_device(pkg):
    """sdpa_host_declare_pinned / sdpa_host_forget_pinned only keep a list: they work without a GPU and reject what
    they cannot mean"""
    import ctypes
    lib = pkg.load()
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.addressof(buf)
    assert lib.sdpa_host_declare_pinned(None, 4096) < 0 and lib.sdpa_host_declare_pinned(p, 0) < 0
    assert lib.sdpa_host_forget_pinned(p) < 0                   # never declared
    assert lib.sdpa_host_declare_pinned(p, 4096) == 0
    assert lib.sdpa_host_declare_pinned(p, 2048) == 0           # declared again: the new size replaces the old one
    assert lib.sdpa_host_forget_pinned(p) == 0
    assert lib.sdpa_host_forget_pinned(p) < 0


def test_argument_validation_precedes_device_use(pkg):
    lib = pkg.load()
    a = np.zeros((4, 4))
    p = a.ctypes.data
    E = pkg._lib.SDPA_EINVAL
    assert lib.sdpa_attention_f64(None, p, p, p, 4, 4, 4, 4, 0) == E
    assert lib.sdpa_attention_f64(p, p, p, p, 0, 4, 4, 4, 0) == E
    assert lib.sdpa_attention_f64(p, p, p, p, 4, 4, -1, 4, 0) == E
    assert lib.sdpa_init(-3) == E
    assert lib.sdpa_dev_cvt_d2f(p, p, 4, 4, 6, None) == E        # ld not a multiple of 4
    assert lib.sdpa_dev_cvt_d2f(p, p, 4, 8, 4, None) == E        # ld < cols
    assert lib.sdpa_dev_shard_partial_f32(p, 4, p, 4, p, 4, p, 4, p, p, 0, 4, 4, 4, None, 0, None) == E
    assert lib.sdpa_dev_shard_partial_f32(p, 4, p, 4, p, 4, p, 3, p, p, 4, 4, 4, 4, None, 0, None) == E
    assert lib.sdpa_dev_shard_partial_f32(p + 4, 4, p, 4, p, 4, p, 4, p, p, 2, 2, 4, 4, None, 0, None) == E  # misaligned
    assert lib.sdpa_owner_count(5, 0, 0) == E
    assert lib.sdpa_dev_kv_splits(8192, 8192, 128, 128) == 8
    assert lib.sdpa_dev_kv_splits(32768, 65536, 128, 128) == 2
    assert lib.sdpa_dev_kv_splits(512, 512, 64, 64) == 4
    assert lib.sdpa_dev_kv_splits(8192, 8192, 512, 512) == 2      # dk-split kernel: 128 workgroups of 64 rows
    assert lib.sdpa_dev_kv_splits(8192, 8192, 600, 64) == 1       # dk > 512: VALU kernel, no splits
    # 2 slabs on a whole chip; the scratch also covers a CU-masked stream, where stream-K cuts a query block into up to 3
    assert lib.sdpa_dev_workspace_bytes(32768, 65536, 128, 128) == 3 * 32768 * 130 * 4 + 256 * 8   # + one arrival word per query block
    assert lib.sdpa_dev_workspace_bytes(100000, 64, 128, 128) == 0


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_never_falls_back(pkg):
    lib = pkg.load()
    assert lib.sdpa_device_count() <= 0
    assert lib.sdpa_init(0) == pkg._lib.SDPA_ENODEV
    Q = np.random.rand(4, 8); K = np.random.rand(6, 8); V = np.random.rand(6, 8)
    with pytest.raises(pkg.SdpaError) as e:
        pkg.attention(Q, K, V)
    assert e.value.code == pkg._lib.SDPA_ENODEV
    with pytest.raises(pkg.SdpaError):
        pkg.HipBackend()
    buf = np.zeros(64, dtype=np.float32).ctypes.data
    assert lib.sdpa_dev_shard_partial_f32(buf, 4, buf, 4, buf, 4, buf, 4, buf, buf, 2, 2, 4, 4,
                                          None, 0, None) == pkg._lib.SDPA_ENODEV


def test_hand_declared_rccl_abi_matches_the_installed_header(tmp_path):
    """sdpa_coll.hip binds RCCL with dlsym() through function-pointer types declared by hand (csrc/sdpa_rccl_abi.h: the
    library does not link against librccl).  csrc/sdpa_rccl_abi_check.cpp static_asserts the three enum values it uses and
    the calling-convention class of every argument of the nine entry points against <rccl/rccl.h>; `make` compiles it
    (-fsyntax-only) with every build.  Here: the check passes on the shipped declarations, and FAILS on a drifted copy
    (a wrong signature would be undefined behaviour before the run-time self-test ever saw it)."""
    hipcc, rccl_h = "/opt/rocm/bin/hipcc", "/opt/rocm/include/rccl/rccl.h"
    if not (os.path.exists(hipcc) and os.path.exists(rccl_h)):
        pytest.skip("needs hipcc and the RCCL header")
    csrc = os.path.join(ROOT, PKG, "csrc")
    cmd = [hipcc, "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-I/opt/rocm/include"]
    r = subprocess.run(cmd + ["-I" + csrc, os.path.join(csrc, "sdpa_rccl_abi_check.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # a drifted declaration: ncclReduce without its `root` argument, ncclMax = 3
    hdr = open(os.path.join(csrc, "sdpa_rccl_abi.h")).read()
    bad = hdr.replace("(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t)",
                      "(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t)").replace("kNcclMax = 2", "kNcclMax = 3")
    assert bad != hdr
    (tmp_path / "sdpa_rccl_abi.h").write_text(bad)
    (tmp_path / "sdpa_rccl_abi_check.cpp").write_text(open(os.path.join(csrc, "sdpa_rccl_abi_check.cpp")).read())
    r = subprocess.run(cmd + ["-I" + str(tmp_path), str(tmp_path / "sdpa_rccl_abi_check.cpp")], capture_output=True, text=True)
    assert r.returncode != 0 and "ncclReduce does not match" in r.stderr and "ncclRedOp_t values" in r.stderr, r.stderr[-2000:]
    # and sdpa_coll.hip uses exactly those declarations (no second copy to drift)
    coll = open(os.path.join(csrc, "sdpa_coll.hip")).read()
    assert '#include "sdpa_rccl_abi.h"' in coll and "typedef struct ncclComm" not in coll


def test_product_never_touches_the_oracle():
    """the product tree must not import, link or execute anything under oracle/"""
    pdir = os.path.join(ROOT, PKG)
    for dirpath, _, files in os.walk(pdir):
        if os.path.basename(dirpath) in ("build", "lib", "bin", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for line in text.splitlines():
                    s = line.strip()
                    if s.startswith(("#include", "import ", "from ")) or "dlopen" in s or "CDLL" in s:
                        assert "oracle" not in s, "%s: %s" % (f, s)
    out = subprocess.run(["ldd", os.path.join(pdir, "lib", "libsdpa_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "libamdhip64" in out


# ---------------------------------------------------------------- CLI host (plain C) ----------
CLI = os.path.join(ROOT, PKG, "bin", "attention-hip")


def _run(args, **kw):
    return subprocess.run([CLI] + args, capture_output=True, text=True, **kw)


def test_cli_usage_and_io_errors(tmp_path):
    """attention.c:165-168 / :86-89 / :103-114 messages and exit codes"""
    r = _run([])
    assert r.returncode == 1 and r.stdout == "" and r.stderr == "Usage: %s <testing data>\n" % CLI
    r = _run(["/no/such/file"])
    assert r.returncode == 1 and r.stderr == "Cannot open file: /no/such/file\n" and r.stdout == ""
    short = tmp_path / "short.bin"
    short.write_bytes(open(os.path.join(ROOT, "tests", "golden", "tiny_D1.bin"), "rb").read()[:200])
    r = _run([str(short)])
    assert r.returncode == 1 and r.stderr == "Invalid testing data.\n" and r.stdout == ""
    hdr = tmp_path / "hdr.bin"
    hdr.write_bytes(b"\x01\x00\x00")
    r = _run([str(hdr)])
    assert r.returncode == 1 and r.stderr == "Invalid testing data.\n"


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_cli_without_gpu_exits_nonzero_and_prints_nothing_on_stdout():
    r = _run([os.path.join(ROOT, "tests", "golden", "tiny_D1.bin")])
    assert r.returncode == 1 and r.stdout == "" and "no usable HIP device" in r.stderr
