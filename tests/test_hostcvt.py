"""CPU: the host-side converter of $SDPA_HOST_CVT=1 (csrc/sdpa_hostcvt.cpp, C ABI sdpa_host_cvt_rows) --
the reference's own placement of the fp64 -> fp32 converts (cvt_d2f_avx512, attention-mpi.c:31-64,
called at :224-225 and :303), here writing the operand images the GPU kernels read.  Its rows must be
the device converters' rows bit for bit: fp32 = round-to-nearest-even of the double (numpy's astype),
bf16 = that float rounded to nearest even on its bit pattern; AVX-512 rows == plain-C rows."""
import ctypes

import numpy as np
import pytest


def cvt(pkg, src, ld, kind, mult=1.0, scalar=False):
    rows, cols = src.shape
    dst = np.full((rows, ld), 0x7f, dtype=np.float32 if kind == 0 else np.uint16)
    src = np.ascontiguousarray(src, dtype=np.float64)
    rc = pkg.load().sdpa_host_cvt_rows(src.ctypes.data, dst.ctypes.data, rows, cols, ld, kind, mult, 1 if scalar else 0)
    assert rc == 0
    return dst


def bf16_bits(x64, mult):
    with np.errstate(over="ignore"):
        f = (x64 * mult).astype(np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffffffff
    return (u >> 16).astype(np.uint16)


def samples(rows, cols, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((rows, cols)) * np.exp(rng.uniform(-30, 30, (rows, cols)))
    flat = x.reshape(-1)
    # exact ties of the fp64 -> fp32 rounding (23-bit mantissa + half an ulp), both parities
    base = np.float32(1.0) + np.arange(8, dtype=np.float32) * np.float32(2.0 ** -23)
    ties = base.astype(np.float64) + 2.0 ** -24
    # ties of the fp32 -> bf16 rounding: 1 + (2k+1) * 2^-8
    ties16 = 1.0 + (2 * np.arange(8) + 1) * 2.0 ** -8
    special = np.concatenate([ties, -ties, ties16, -ties16, [0.0, -0.0, 1e-320, -1e-45, 3.4028235677973366e38, 1e39, -1e39,
                                                              65504.0, 2.0 ** -126, 2.0 ** -149, 1.0 + 2.0 ** -52]])
    flat[:min(len(special), flat.size)] = special[:flat.size]
    return x


@pytest.mark.parametrize("cols,ld", [(1, 4), (7, 8), (8, 8), (9, 12), (64, 64), (72, 128), (100, 128), (128, 128), (300, 300), (513, 516)])
def test_fp32_rows_are_numpy_astype_rows(cols, ld, pkg):
    x = samples(37, cols, cols)
    with np.errstate(over="ignore"):
        want = x.astype(np.float32)
    for scalar in (False, True):
        got = cvt(pkg, x, ld, 0, scalar=scalar)
        assert np.array_equal(got[:, :cols].view(np.uint32), want.view(np.uint32)), "scalar=%s" % scalar
        assert not got[:, cols:].any(), "pad columns must be zero"


@pytest.mark.parametrize("cols,ld", [(1, 64), (7, 64), (8, 8), (64, 64), (100, 128), (250, 256), (512, 512), (13, 13)])
@pytest.mark.parametrize("mult", [1.0, float(np.float32(1.44269504088896340736) * (np.float32(1.0) / np.sqrt(np.float32(128.0))))])
def test_bf16_rows_are_the_device_converters_two_roundings(cols, ld, mult, pkg):
    x = samples(29, cols, 1000 + cols)
    want = bf16_bits(x, mult)
    for scalar in (False, True):
        got = cvt(pkg, x, ld, 1, mult, scalar=scalar)
        assert np.array_equal(got[:, :cols], want), "scalar=%s" % scalar
        assert not got[:, cols:].any()


def _aligned(rows, ld, dtype, offset_bytes=0):
    """a (rows, ld) array whose first byte sits `offset_bytes` behind a 64-byte boundary"""
    item = np.dtype(dtype).itemsize
    raw = np.full(rows * ld * item + 128, 0x7f, dtype=np.uint8)
    start = (-raw.ctypes.data) % 64 + offset_bytes
    return raw[start:start + rows * ld * item].view(dtype).reshape(rows, ld)


@pytest.mark.parametrize("kind,cols,ld", [(0, 128, 128), (0, 512, 512), (0, 100, 128), (0, 72, 128), (0, 16, 16), (0, 300, 304), (0, 9, 12),
                                          (1, 128, 128), (1, 512, 512), (1, 100, 128), (1, 250, 256), (1, 32, 32), (1, 13, 16)])
@pytest.mark.parametrize("offset", [0, 16, 4])
def test_streaming_store_rows_write_the_same_bytes(kind, cols, ld, offset, pkg):
    """flags bit 1: streaming (non-temporal) stores for line-aligned destination rows -- whole 64-byte lines of a row,
    ordinary stores for unaligned rows, row tails and pad columns: the SAME image as the ordinary rows, at any
    destination alignment (rows of a 64-byte multiple stay aligned, others alternate)"""
    lib = pkg.load()
    x = np.ascontiguousarray(samples(41, cols, 7 * cols + kind))
    mult = 1.0 if kind == 0 else 0.1275
    dt = np.float32 if kind == 0 else np.uint16
    if offset % np.dtype(dt).itemsize:
        pytest.skip("not an element boundary")
    outs = []
    for flags in (4, 2, 1):                                   # ordinary stores, streaming stores, plain C
        dst = _aligned(41, ld, dt, offset)
        assert lib.sdpa_host_cvt_rows(x.ctypes.data, dst.ctypes.data, 41, cols, ld, kind, mult, flags) == 0
        outs.append(dst.copy())
    assert np.array_equal(outs[0].view(np.uint8), outs[1].view(np.uint8))
    assert np.array_equal(outs[0].view(np.uint8), outs[2].view(np.uint8))
    assert not outs[1][:, cols:].any()


def test_argument_checks(pkg):
    lib = pkg.load()
    buf = (ctypes.c_double * 16)()
    p = ctypes.addressof(buf)
    assert lib.sdpa_host_cvt_rows(p, p, 1, 8, 4, 0, 1.0, 0) == pkg._lib.SDPA_EINVAL      # ld < cols
    assert lib.sdpa_host_cvt_rows(p, p, 1, 4, 4, 3, 1.0, 0) == pkg._lib.SDPA_EINVAL      # unknown kind
    assert lib.sdpa_host_cvt_rows(p, p, 1, 4, 4, 2, 1.0, 0) == pkg._lib.SDPA_EINVAL      # kind 2: ld must be the padded dk
    assert lib.sdpa_host_cvt_rows(None, p, 1, 4, 4, 0, 1.0, 0) == pkg._lib.SDPA_EINVAL
    assert lib.sdpa_host_cvt_rows(None, None, 0, 4, 4, 0, 1.0, 0) == 0


# ---- the host-side widening of result rows ($SDPA_HOST_WIDEN; cvt_f2d_avx512, attention-mpi.c:68-101) ----
@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 1000, 16384 * 3 + 5, 16384 * 40 + 1])
@pytest.mark.parametrize("threads", [1, 4])
def test_host_widen_is_exact_for_every_length_alignment_and_thread_count(pkg, n, threads):
    rng = np.random.default_rng(n + threads)
    src = rng.standard_normal(n + 3).astype(np.float32)
    if n > 4:
        src[1:5] = [np.inf, -0.0, np.float32(1e-45), -np.inf]          # specials and a subnormal survive
    for off in (0, 1, 3):                                             # dst 64-byte aligned or not
        dst = np.full(n + 16, -7.0)
        lib = pkg.load()
        rc = lib.sdpa_host_widen(src[off:].ctypes.data, dst[off:].ctypes.data, n, threads, 0)
        assert rc == 0
        assert np.array_equal(dst[off:off + n], src[off:off + n].astype(np.float64))
        assert (dst[:off] == -7.0).all() and (dst[off + n:] == -7.0).all(), "wrote outside its range"
    dst2 = np.empty(n)
    assert pkg.load().sdpa_host_widen(src.ctypes.data, dst2.ctypes.data, n, 1, 1) == 0      # the plain C loop
    assert np.array_equal(dst2, src[:n].astype(np.float64))


def test_host_widen_argument_errors(pkg):
    lib = pkg.load()
    a = np.zeros(4, np.float32)
    assert lib.sdpa_host_widen(None, a.ctypes.data, 4, 1, 0) == pkg._lib.SDPA_EINVAL
    assert lib.sdpa_host_widen(a.ctypes.data, None, 4, 1, 0) == pkg._lib.SDPA_EINVAL
    assert lib.sdpa_host_widen(None, None, 0, 1, 0) == 0


# ---- the transposed V image of the bf16 kernels, made on the host (sdpa_host_cvt_vt; round 5) ----------------------------------
def _kvpos(j):
    return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1)


def _vt_reference(V, keys_pad, cols_pad, ldt):
    """cvt_d2bf_t_kernel restated in numpy: dst[c, kvpos(j)] = bf16(float(V[j, c])), RNE twice, zeros elsewhere"""
    keys, cols = V.shape
    f = V.astype(np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    b = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)          # (finite inputs: no carry out of 32 bits matters)
    img = np.zeros((cols_pad, ldt), np.uint16)
    pos = np.array([(j & ~31) | _kvpos(j & 31) for j in range(keys)], dtype=np.int64)
    img[:cols, pos] = b.T
    return img


@pytest.mark.parametrize("keys,cols,cols_pad,extra_tiles,scalar", [
    (64, 32, 32, 0, False), (96, 256, 256, 0, False), (33, 200, 256, 0, False), (1, 7, 64, 1, False), (255, 129, 256, 2, False),
    (64, 40, 64, 0, True), (95, 256, 256, 0, True), (0, 16, 32, 1, False)])
def test_host_vt_image_is_the_device_converters_image(keys, cols, cols_pad, extra_tiles, scalar, pkg):
    """every key position of the padded range is written (stale staging bytes must not survive), the pad rows are zero, the values
    and their places are cvt_d2bf_t_kernel's; AVX-512 rows + streaming lines == the plain C rows"""
    lib = pkg.load()
    rng = np.random.default_rng(keys * 1000 + cols)
    V = rng.uniform(-3, 3, (keys, cols))
    if keys:
        V[0, 0] = 1.0 + 2.0 ** -9                        # an RNE tie of the fp32 -> bf16 rounding (to even: down)
        V[-1, -1] = -(1.0 + 3 * 2.0 ** -9)               # ... and one that rounds up
    keys_pad = (keys + 31) // 32 * 32 + 32 * extra_tiles
    ldt = keys_pad + 64                                  # the image is wider than the entry: the neighbours must stay untouched
    img = np.full((cols_pad, ldt), 0xABCD, np.uint16)
    rc = lib.sdpa_host_cvt_vt(V.ctypes.data if keys else None, img.ctypes.data, keys, keys_pad, cols, cols_pad, ldt, 1, 1 if scalar else 0)
    assert rc == 0
    want = _vt_reference(V, keys_pad, cols_pad, ldt)
    assert np.array_equal(img[:, :keys_pad], want[:, :keys_pad])
    assert (img[:, keys_pad:] == 0xABCD).all()


def test_host_vt_entries_tile_by_tile_equal_one_call(pkg):
    """an entry converted 64 keys at a time at its offsets inside a larger image equals one call over the whole range -- what the
    converter pool's work items do (the pool itself runs inside the engine's streamed bf16 call: tests/test_gpu_host_pipeline.py)"""
    lib = pkg.load()
    rng = np.random.default_rng(5)
    keys, cols = 200, 96
    V = rng.uniform(-1, 1, (keys, cols))
    keys_pad, ldt = 224, 512
    whole = np.zeros((128, ldt), np.uint16)
    assert lib.sdpa_host_cvt_vt(V.ctypes.data, whole.ctypes.data, keys, keys_pad, cols, 128, ldt, 1, 0) == 0
    parts = np.zeros((128, ldt), np.uint16)
    for r0 in range(0, keys_pad, 64):
        left = max(0, min(64, keys - r0))
        src = V[r0:r0 + left] if left else V[:0]
        src = np.ascontiguousarray(src)
        assert lib.sdpa_host_cvt_vt(src.ctypes.data if left else None, parts.ctypes.data + 2 * r0, left, 64 if r0 + 64 <= keys_pad else keys_pad - r0,
                                    cols, 128, ldt, 1, 0) == 0
    assert np.array_equal(whole, parts)


def test_host_vt_rejects_bad_arguments(pkg):
    lib = pkg.load()
    x = np.zeros(64)
    d = np.zeros(4096, np.uint16)
    assert lib.sdpa_host_cvt_vt(x.ctypes.data, d.ctypes.data, 8, 30, 8, 8, 64, 1, 0) == pkg._lib.SDPA_EINVAL     # keys_pad not whole tiles
    assert lib.sdpa_host_cvt_vt(x.ctypes.data, d.ctypes.data, 8, 32, 8, 4, 64, 1, 0) == pkg._lib.SDPA_EINVAL     # cols_pad < cols
    assert lib.sdpa_host_cvt_vt(x.ctypes.data, d.ctypes.data, 8, 32, 8, 8, 16, 1, 0) == pkg._lib.SDPA_EINVAL     # ldt < keys_pad
    assert lib.sdpa_host_cvt_vt(None, d.ctypes.data, 8, 32, 8, 8, 64, 1, 0) == pkg._lib.SDPA_EINVAL


@pytest.mark.parametrize("keys,cols,cols_pad,threads,item_kb", [(4096, 256, 256, 8, 64), (1000, 200, 256, 3, 4), (8192, 64, 64, 5, 1024)])
def test_host_vt_on_a_pool_of_threads_is_the_single_thread_image(keys, cols, cols_pad, threads, item_kb, pkg, monkeypatch):
    """the pool's work items (whole 32-key tiles, $SDPA_DEBUG=host_cvt_item_kb of source each, taken by whichever thread is free) write the
    same image as one thread does, including the zero tail and the pad rows"""
    lib = pkg.load()
    monkeypatch.setenv("SDPA_DEBUG", "host_cvt_item_kb=%d" % item_kb)
    V = np.random.default_rng(keys + threads).normal(0, 2, (keys, cols))
    keys_pad = (keys + 31) // 32 * 32 + 64
    one = np.full((cols_pad, keys_pad), 0x1234, np.uint16)
    many = np.full((cols_pad, keys_pad), 0x4321, np.uint16)
    assert lib.sdpa_host_cvt_vt(V.ctypes.data, one.ctypes.data, keys, keys_pad, cols, cols_pad, keys_pad, 1, 0) == 0
    assert lib.sdpa_host_cvt_vt(V.ctypes.data, many.ctypes.data, keys, keys_pad, cols, cols_pad, keys_pad, threads, 0) == 0
    assert np.array_equal(one, many)
    assert np.array_equal(one, _vt_reference(V, keys_pad, cols_pad, keys_pad))


def test_host_vt_image_fuzz(pkg):
    """seeded random shapes (keys, cols, pad rows, extra zero tiles, row stride, thread count, item size): the image is the numpy
    restatement's, everything outside [0, keys_pad) x [0, cols_pad) is untouched -- 60 cases, special values included"""
    import os
    lib = pkg.load()
    rng = np.random.default_rng(20260922)
    old = os.environ.get("SDPA_DEBUG")
    try:
        for case in range(60):
            keys = int(rng.integers(0, 700))
            cols = int(rng.integers(1, 200))
            cols_pad = cols + int(rng.integers(0, 70))
            keys_pad = (keys + 31) // 32 * 32 + 32 * int(rng.integers(0, 3))
            if keys_pad == 0:
                keys_pad = 32
            ldt = keys_pad + 32 * int(rng.integers(0, 3))
            threads = int(rng.choice([1, 1, 2, 5]))
            os.environ["SDPA_DEBUG"] = "host_cvt_item_kb=%d" % int(rng.choice([4, 64, 300]))
            V = rng.normal(0, 3, (keys, cols))
            if keys:
                V.flat[rng.integers(0, V.size, min(V.size, 6))] = [0.0, -0.0, 1e-45, -3.0e38, 65504.0, 1.0 + 2.0 ** -9][:min(V.size, 6)]
            img = np.full((cols_pad + 1, ldt), 0x5A5A, np.uint16)        # (+1 row: the converter must not write behind cols_pad)
            rc = lib.sdpa_host_cvt_vt(V.ctypes.data if keys else None, img.ctypes.data, keys, keys_pad, cols, cols_pad, ldt, threads,
                                      int(rng.choice([0, 1, 2, 4])))
            assert rc == 0, (case, keys, cols)
            want = _vt_reference(V, keys_pad, cols_pad, ldt)
            assert np.array_equal(img[:cols_pad, :keys_pad], want[:, :keys_pad]), (case, keys, cols, cols_pad, keys_pad, ldt, threads)
            assert (img[:cols_pad, keys_pad:] == 0x5A5A).all() and (img[cols_pad] == 0x5A5A).all(), (case, "wrote outside its range")
    finally:
        if old is None:
            os.environ.pop("SDPA_DEBUG", None)
        else:
            os.environ["SDPA_DEBUG"] = old


# ---- the TILED images of dv > 256 (round 6: include/sdpa_hip.h) -----------------------------------------------------------------
def _bf16(x, mult=1.0):
    return bf16_bits(np.asarray(x, dtype=np.float64), mult)


def _tiled_k_reference(K, ld):
    n, dk = K.shape
    b = np.zeros((n, ld), np.uint16)
    b[:, :dk] = _bf16(K)
    swz = min(15, ld // 8 - 1)
    out = np.empty_like(b)
    for r in range(n):
        out[r] = b[r].reshape(-1, 8)[np.arange(ld // 8) ^ (r & swz)].reshape(-1)
    return out


def _tiled_vt_reference(V, keys_pad, cols_pad):
    keys, cols = V.shape
    b = np.zeros((keys_pad, cols_pad), np.uint16)
    b[:keys, :cols] = _bf16(V)
    img = np.zeros((keys_pad // 32, cols_pad // 512, 512, 32), np.uint16)
    col = np.arange(cols_pad)
    x = (col >> 2) & 3
    for t in range(keys_pad // 32):
        line = np.zeros((cols_pad, 32), np.uint16)
        for j in range(32):
            line[:, _kvpos(j)] = b[32 * t + j]
        line = line.reshape(cols_pad, 4, 8)
        out = np.empty_like(line)
        for q in range(4):
            out[col, q] = line[col, q ^ x]
        img[t] = out.reshape(cols_pad // 512, 512, 32)
    return img.reshape(-1)


@pytest.mark.parametrize("rows,cols,ld", [(64, 512, 512), (37, 300, 512), (50, 256, 256), (33, 100, 128), (40, 64, 64), (19, 7, 64)])
def test_tiled_k_rows_are_the_chunk_swizzled_bf16_rows(rows, cols, ld, pkg):
    lib = pkg.load()
    x = np.ascontiguousarray(samples(rows, cols, 31 * cols + rows))
    want = _tiled_k_reference(x, ld)
    for flags in (0, 1, 2, 4):                      # AVX-512 (default stores), plain C, streaming stores on / off
        dst = _aligned(rows, ld, np.uint16, 0)
        assert lib.sdpa_host_cvt_rows(x.ctypes.data, dst.ctypes.data, rows, cols, ld, 2, 1.0, flags) == 0
        assert np.array_equal(dst, want), "flags=%d" % flags


@pytest.mark.parametrize("keys,cols,cols_pad,extra_tiles,scalar", [
    (96, 512, 512, 0, False), (33, 300, 512, 0, False), (95, 512, 512, 1, True), (64, 700, 1024, 0, False), (1, 257, 512, 2, False),
    (0, 400, 512, 1, False), (130, 1024, 1024, 0, True)])
def test_host_tiled_vt_image_is_the_documented_layout(keys, cols, cols_pad, extra_tiles, scalar, pkg):
    lib = pkg.load()
    V = np.random.default_rng(keys * 7 + cols).uniform(-3, 3, (keys, cols))
    keys_pad = (keys + 31) // 32 * 32 + 32 * extra_tiles
    if keys_pad == 0:
        keys_pad = 32
    img = np.full(keys_pad * cols_pad + 64, 0xABCD, np.uint16)
    rc = lib.sdpa_host_cvt_vt(V.ctypes.data if keys else None, img.ctypes.data, keys, keys_pad, cols, cols_pad, keys_pad, 1, 1 if scalar else 0)
    assert rc == 0
    assert np.array_equal(img[:keys_pad * cols_pad], _tiled_vt_reference(V, keys_pad, cols_pad))
    assert (img[keys_pad * cols_pad:] == 0xABCD).all(), "wrote behind the image"


@pytest.mark.parametrize("keys,cols,cols_pad,threads,item_kb", [(4096, 512, 512, 8, 64), (1000, 300, 512, 3, 4), (2048, 1000, 1024, 5, 256)])
def test_host_tiled_vt_on_a_pool_of_threads_is_the_single_thread_image(keys, cols, cols_pad, threads, item_kb, pkg, monkeypatch):
    lib = pkg.load()
    monkeypatch.setenv("SDPA_DEBUG", "host_cvt_item_kb=%d" % item_kb)
    V = np.random.default_rng(keys + threads).normal(0, 2, (keys, cols))
    keys_pad = (keys + 31) // 32 * 32 + 64
    one = np.full(cols_pad * keys_pad, 0x1234, np.uint16)
    many = np.full(cols_pad * keys_pad, 0x4321, np.uint16)
    assert lib.sdpa_host_cvt_vt(V.ctypes.data, one.ctypes.data, keys, keys_pad, cols, cols_pad, keys_pad, 1, 0) == 0
    assert lib.sdpa_host_cvt_vt(V.ctypes.data, many.ctypes.data, keys, keys_pad, cols, cols_pad, keys_pad, threads, 0) == 0
    assert np.array_equal(one, many)
    assert np.array_equal(one, _tiled_vt_reference(V, keys_pad, cols_pad))


def test_host_tiled_vt_rejects_a_partial_chunk(pkg):
    lib = pkg.load()
    x = np.zeros(64 * 300)
    d = np.zeros(32 * 512, np.uint16)
    assert lib.sdpa_host_cvt_vt(x.ctypes.data, d.ctypes.data, 32, 32, 300, 300, 32, 1, 0) == pkg._lib.SDPA_EINVAL     # cols_pad must be whole 512-column chunks
