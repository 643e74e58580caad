// sdpa_hostcvt.cpp -- see sdpa_hostcvt.h.  Host code only: compiled by g++ (x86-64), no device pass.
#include "sdpa_hostcvt.h"
#include "sdpa_debug.h"

#include <hip/hip_runtime_api.h>
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <thread>
#include <vector>

namespace sdpa {
namespace {

inline void relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

// (float)double under the default rounding mode is round-to-nearest-even: what __double2float_rn and
// _mm512_cvtpd_ps (attention-mpi.c:31-64) do.
void rows_to_f32_scalar(const double *src, float *dst, long rows, int cols, int ld) {
    for (long r = 0; r < rows; ++r) {
        const double *s = src + r * cols;
        float *d = dst + r * ld;
        for (int c = 0; c < cols; ++c) d[c] = (float)s[c];
        for (int c = cols; c < ld; ++c) d[c] = 0.f;
    }
}

// the device converter's two roundings: fp64 -> fp32 (RNE), fp32 -> bf16 (RNE on the bit pattern)
inline unsigned short to_bf16(double x) {
    const float f = (float)x;
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

void rows_to_bf16_scalar(const double *src, unsigned short *dst, long rows, int cols, int ld, double mult) {
    for (long r = 0; r < rows; ++r) {
        const double *s = src + r * cols;
        unsigned short *d = dst + r * ld;
        for (int c = 0; c < cols; ++c) d[c] = to_bf16(s[c] * mult);       // (x * 1.0 is x)
        for (int c = cols; c < ld; ++c) d[c] = 0;
    }
}

// AVX-512 rows: 8 doubles per vcvtpd2ps, as the reference's cvt_d2f_avx512 (attention-mpi.c:31-64); the
// bf16 form does the device kernel's integer rounding on 8 lanes at a time.  Masked tail, no alignment
// assumed.  Same results as the scalar rows bit for bit (tests/test_hostcvt.py).
__attribute__((target("avx512f,avx512bw,avx512vl")))
void rows_to_f32_avx512(const double *src, float *dst, long rows, int cols, int ld) {
    for (long r = 0; r < rows; ++r) {
        const double *s = src + r * cols;
        float *d = dst + r * ld;
        int c = 0;
        for (; c + 8 <= cols; c += 8) _mm256_storeu_ps(d + c, _mm512_cvtpd_ps(_mm512_loadu_pd(s + c)));
        if (c < cols) {
            const __mmask8 k = (__mmask8)((1u << (cols - c)) - 1u);
            _mm256_mask_storeu_ps(d + c, k, _mm512_cvtpd_ps(_mm512_maskz_loadu_pd(k, s + c)));
        }
        for (c = cols; c < ld; ++c) d[c] = 0.f;
    }
}

__attribute__((target("avx512f,avx512bw,avx512vl")))
void rows_to_bf16_avx512(const double *src, unsigned short *dst, long rows, int cols, int ld, double mult) {
    const __m512d vm = _mm512_set1_pd(mult);
    const __m256i one = _mm256_set1_epi32(1), bias = _mm256_set1_epi32(0x7fff);
    for (long r = 0; r < rows; ++r) {
        const double *s = src + r * cols;
        unsigned short *d = dst + r * ld;
        int c = 0;
        for (; c < cols; c += 8) {
            const int left = cols - c;
            const __mmask8 k = left >= 8 ? (__mmask8)0xff : (__mmask8)((1u << left) - 1u);
            const __m512d x = _mm512_mul_pd(_mm512_maskz_loadu_pd(k, s + c), vm);
            __m256i u = _mm256_castps_si256(_mm512_cvtpd_ps(x));
            u = _mm256_add_epi32(u, _mm256_add_epi32(bias, _mm256_and_si256(_mm256_srli_epi32(u, 16), one)));
            _mm_mask_storeu_epi16(d + c, k, _mm256_cvtepi32_epi16(_mm256_srli_epi32(u, 16)));
        }
        for (c = cols; c < ld; ++c) d[c] = 0;
    }
}

// Rows of the TILED K image (round 6, sdpa_internal.h): the same values, 16-byte chunk c of image row r stored at chunk position
// c ^ (r & swz) -- the byte order of the tandem kernel's LDS K buffer.  `row0` = image row of the first row (the swizzle needs it).
void rows_to_bf16_swz_scalar(const double *src, unsigned short *dst, long rows, int cols, int ld, double mult, long row0, int swz) {
    for (long r = 0; r < rows; ++r) {
        const double *s = src + r * cols;
        unsigned short *d = dst + r * ld;
        const int x = (int)((row0 + r) & swz);
        for (int c = 0; c < ld; ++c) d[(((c >> 3) ^ x) << 3) | (c & 7)] = c < cols ? to_bf16(s[c] * mult) : (unsigned short)0;
    }
}

// AVX-512: a row in 64-byte lines of four chunks; x & 3 permutes the chunks of a line, x >> 2 the lines (ld is a multiple of 32)
__attribute__((target("avx512f,avx512bw,avx512vl")))
void rows_to_bf16_swz_avx512(const double *src, unsigned short *dst, long rows, int cols, int ld, double mult, long row0, int swz, bool nt) {
    const __m512d vm = _mm512_set1_pd(mult);
    const __m256i one = _mm256_set1_epi32(1), bias = _mm256_set1_epi32(0x7fff);
    for (long r = 0; r < rows; ++r) {
        const double *s = src + r * cols;
        unsigned short *d = dst + r * ld;
        const int x = (int)((row0 + r) & swz), xl = x >> 2, xc = x & 3;
        const bool aligned = nt && (((uintptr_t)d) & 63u) == 0;
        for (int line = 0; line < ld / 32; ++line) {
            __m128i q[4];
            for (int k = 0; k < 4; ++k) {
                const int c = line * 32 + k * 8, left = cols - c;
                const __mmask8 msk = left >= 8 ? (__mmask8)0xff : left <= 0 ? (__mmask8)0 : (__mmask8)((1u << left) - 1u);
                __m256i u = _mm256_castps_si256(_mm512_cvtpd_ps(_mm512_mul_pd(_mm512_maskz_loadu_pd(msk, s + (left > 0 ? c : 0)), vm)));
                u = _mm256_add_epi32(u, _mm256_add_epi32(bias, _mm256_and_si256(_mm256_srli_epi32(u, 16), one)));
                q[k] = _mm_maskz_mov_epi16(msk, _mm256_cvtepi32_epi16(_mm256_srli_epi32(u, 16)));      // (pad columns: zero, not bf16(0 + bias))
            }
            __m512i v = _mm512_castsi128_si512(q[0 ^ xc]);
            v = _mm512_inserti32x4(v, q[1 ^ xc], 1);
            v = _mm512_inserti32x4(v, q[2 ^ xc], 2);
            v = _mm512_inserti32x4(v, q[3 ^ xc], 3);
            unsigned short *at = d + (size_t)(line ^ xl) * 32;
            if (aligned) _mm512_stream_si512((__m512i *)at, v);
            else _mm512_storeu_si512((void *)at, v);
        }
    }
    if (nt) _mm_sfence();
}

// The same rows with STREAMING stores (round 4): a row whose destination is 64-byte aligned is written in whole cache
// lines past the cache (two vcvtpd2ps per line of floats, four 8-lane groups per line of bf16), so that the write
// does not first READ the line it overwrites (write-allocate: 16 instead of 12 bytes of memory traffic per fp32
// element) -- the staging image is read next by the copy engine, never by this core.  Unaligned rows, row tails and pad
// columns take the ordinary stores.  The caller fences (sfence) before it publishes the rows.
__attribute__((target("avx512f,avx512bw,avx512vl")))
void rows_to_f32_avx512_nt(const double *src, float *dst, long rows, int cols, int ld) {
    for (long r = 0; r < rows; ++r) {
        const double *s = src + r * cols;
        float *d = dst + r * ld;
        int c = 0;
        if (((uintptr_t)d & 63u) == 0)
            for (; c + 16 <= cols; c += 16) {
                const __m256 lo = _mm512_cvtpd_ps(_mm512_loadu_pd(s + c)), hi = _mm512_cvtpd_ps(_mm512_loadu_pd(s + c + 8));
                const __m512d both = _mm512_insertf64x4(_mm512_castpd256_pd512(_mm256_castps_pd(lo)), _mm256_castps_pd(hi), 1);
                _mm512_stream_ps(d + c, _mm512_castpd_ps(both));
            }
        for (; c + 8 <= cols; c += 8) _mm256_storeu_ps(d + c, _mm512_cvtpd_ps(_mm512_loadu_pd(s + c)));
        if (c < cols) {
            const __mmask8 k = (__mmask8)((1u << (cols - c)) - 1u);
            _mm256_mask_storeu_ps(d + c, k, _mm512_cvtpd_ps(_mm512_maskz_loadu_pd(k, s + c)));
        }
        for (c = cols; c < ld; ++c) d[c] = 0.f;
    }
    _mm_sfence();
}

__attribute__((target("avx512f,avx512bw,avx512vl")))
void rows_to_bf16_avx512_nt(const double *src, unsigned short *dst, long rows, int cols, int ld, double mult) {
    const __m512d vm = _mm512_set1_pd(mult);
    const __m256i one = _mm256_set1_epi32(1), bias = _mm256_set1_epi32(0x7fff);
#define SDPA_BF16X8(P, OUT)                                                                                   \
    do {                                                                                                      \
        __m256i u_ = _mm256_castps_si256(_mm512_cvtpd_ps(_mm512_mul_pd(_mm512_loadu_pd(P), vm)));             \
        u_ = _mm256_add_epi32(u_, _mm256_add_epi32(bias, _mm256_and_si256(_mm256_srli_epi32(u_, 16), one))); \
        OUT = _mm256_cvtepi32_epi16(_mm256_srli_epi32(u_, 16));                                               \
    } while (0)
    for (long r = 0; r < rows; ++r) {
        const double *s = src + r * cols;
        unsigned short *d = dst + r * ld;
        int c = 0;
        if (((uintptr_t)d & 63u) == 0)
            for (; c + 32 <= cols; c += 32) {
                __m128i q0, q1, q2, q3;
                SDPA_BF16X8(s + c, q0);
                SDPA_BF16X8(s + c + 8, q1);
                SDPA_BF16X8(s + c + 16, q2);
                SDPA_BF16X8(s + c + 24, q3);
                __m512i v = _mm512_castsi128_si512(q0);
                v = _mm512_inserti32x4(v, q1, 1);
                v = _mm512_inserti32x4(v, q2, 2);
                v = _mm512_inserti32x4(v, q3, 3);
                _mm512_stream_si512((__m512i *)(d + c), v);
            }
        for (; c < cols; c += 8) {
            const int left = cols - c;
            const __mmask8 k = left >= 8 ? (__mmask8)0xff : (__mmask8)((1u << left) - 1u);
            const __m512d x = _mm512_mul_pd(_mm512_maskz_loadu_pd(k, s + c), vm);
            __m256i u = _mm256_castps_si256(_mm512_cvtpd_ps(x));
            u = _mm256_add_epi32(u, _mm256_add_epi32(bias, _mm256_and_si256(_mm256_srli_epi32(u, 16), one)));
            _mm_mask_storeu_epi16(d + c, k, _mm256_cvtepi32_epi16(_mm256_srli_epi32(u, 16)));
        }
        for (c = cols; c < ld; ++c) d[c] = 0;
    }
    _mm_sfence();
}
#undef SDPA_BF16X8

// $SDPA_DEBUG host_cvt_nt: 1 / 0 = streaming stores in the converter pool on / off (the default: see sdpa_hostcvt.h)
bool stream_stores_default() {
    static const bool on = [] {
        return sdpa_debug_int("host_cvt_nt", SDPA_HOST_CVT_NT_DEFAULT) != 0;
    }();
    return on;
}

bool have_avx512() {
    static const bool yes = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                            __builtin_cpu_supports("avx512vl");
    return yes;
}

__attribute__((target("avx512f"))) inline void stream_line_64(void *dst, const void *src) {
    _mm512_stream_si512((__m512i *)dst, _mm512_load_si512((const __m512i *)src));
}

inline void rows_to_f32(const double *src, float *dst, long rows, int cols, int ld, bool nt) {
    if (!have_avx512()) rows_to_f32_scalar(src, dst, rows, cols, ld);
    else if (nt) rows_to_f32_avx512_nt(src, dst, rows, cols, ld);
    else rows_to_f32_avx512(src, dst, rows, cols, ld);
}

inline void rows_to_bf16(const double *src, unsigned short *dst, long rows, int cols, int ld, double mult, bool nt) {
    if (!have_avx512()) rows_to_bf16_scalar(src, dst, rows, cols, ld, mult);
    else if (nt) rows_to_bf16_avx512_nt(src, dst, rows, cols, ld, mult);
    else rows_to_bf16_avx512(src, dst, rows, cols, ld, mult);
}

inline void rows_to_bf16_swz(const double *src, unsigned short *dst, long rows, int cols, int ld, double mult, long row0, int swz,
                             bool nt, bool force_scalar = false) {
    if (force_scalar || !have_avx512() || (ld & 31)) rows_to_bf16_swz_scalar(src, dst, rows, cols, ld, mult, row0, swz);
    else rows_to_bf16_swz_avx512(src, dst, rows, cols, ld, mult, row0, swz, nt);
}

// The Vt image of the bf16 kernels, made on the host (round 5: what a persistent bf16 launch is fed with -- a device transpose
// kernel could not run beside it).  `keys` rows of V (fp64, `cols` columns) starting at row r0 of an entry, r0 a multiple of 32,
// become columns of the entry's image: dst[c * ldt + r0 + kvpos(j)] = bf16((float)V[r0 + j][c]) for the keys of each 32-key tile,
// zero for tile positions behind the entry's last key (`keys` may end inside a tile: the caller passes whole tiles up to the
// padded key count), zero rows for c in [cols, cols_pad).  kvpos swaps bits 2 and 3 of the key's index (sdpa_internal.h:
// bf16_kvpos -- each 16-key group is stored 0-3, 8-11, 4-7, 12-15).  The same two roundings as everywhere (RNE, RNE): bit for bit
// cvt_d2bf_t_kernel's image.  Per tile: the rows through the row converter into a 32 x cols scratch (L1 / L2), then 32 x 32
// blocks of it transposed into whole 64-byte lines of the image (streaming stores when the line is aligned).
inline int kvpos32(int j) { return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1); }

// 32 x 32 block of 16-bit values, transposed in registers: out line c (64 bytes at out + c * ldt) = in[kvpos(p)][c], p = 0..31.
// Three unpack stages (16, 32, 64 bits: rows pair up, then fours, then eights) leave, per 128-bit lane, eight rows of one column;
// a 4 x 4 transpose of lanes puts a column's 32 rows into one register.  `nc` columns are stored (the scratch's pad columns are
// transposed too, and dropped).
__attribute__((target("avx512f,avx512bw,avx512vl")))
void transpose_block_32x32(const unsigned short *in, int ldin, unsigned short *out, long ldt, int nc, bool nt) {
    __m512i r[32], a[32];
    for (int p = 0; p < 32; ++p) r[p] = _mm512_loadu_si512((const void *)(in + (size_t)kvpos32(p) * ldin));
    for (int i = 0; i < 16; ++i) {
        a[2 * i] = _mm512_unpacklo_epi16(r[2 * i], r[2 * i + 1]);
        a[2 * i + 1] = _mm512_unpackhi_epi16(r[2 * i], r[2 * i + 1]);
    }
    for (int j = 0; j < 8; ++j) {
        r[4 * j] = _mm512_unpacklo_epi32(a[4 * j], a[4 * j + 2]);
        r[4 * j + 1] = _mm512_unpackhi_epi32(a[4 * j], a[4 * j + 2]);
        r[4 * j + 2] = _mm512_unpacklo_epi32(a[4 * j + 1], a[4 * j + 3]);
        r[4 * j + 3] = _mm512_unpackhi_epi32(a[4 * j + 1], a[4 * j + 3]);
    }
    for (int m = 0; m < 4; ++m)
        for (int q = 0; q < 4; ++q) {
            a[8 * m + 2 * q] = _mm512_unpacklo_epi64(r[8 * m + q], r[8 * m + 4 + q]);
            a[8 * m + 2 * q + 1] = _mm512_unpackhi_epi64(r[8 * m + q], r[8 * m + 4 + q]);
        }
    // a[8m + k], lane l = rows 8m .. 8m+7 of column 8l + k
    const bool aligned = nt && (((uintptr_t)out | (uintptr_t)(ldt * 2)) & 63u) == 0;
    for (int k = 0; k < 8; ++k) {
        const __m512i t0 = _mm512_shuffle_i32x4(a[k], a[8 + k], 0x44), t1 = _mm512_shuffle_i32x4(a[k], a[8 + k], 0xEE);
        const __m512i t2 = _mm512_shuffle_i32x4(a[16 + k], a[24 + k], 0x44), t3 = _mm512_shuffle_i32x4(a[16 + k], a[24 + k], 0xEE);
        const __m512i y[4] = {_mm512_shuffle_i32x4(t0, t2, 0x88), _mm512_shuffle_i32x4(t0, t2, 0xDD),
                              _mm512_shuffle_i32x4(t1, t3, 0x88), _mm512_shuffle_i32x4(t1, t3, 0xDD)};
        for (int l = 0; l < 4; ++l) {
            const int c = 8 * l + k;
            if (c >= nc) continue;
            if (aligned) _mm512_stream_si512((__m512i *)(out + (size_t)c * ldt), y[l]);
            else _mm512_storeu_si512((void *)(out + (size_t)c * ldt), y[l]);
        }
    }
}

// this thread's scratch of the transposing converter (a group of rows in bf16 + one column block of its image), grown on demand;
// the pool's workers size and touch it when they start, not inside a caller's timed call
unsigned short *vt_scratch(size_t elems) {
    static thread_local std::vector<unsigned short> store;
    if (store.size() < elems + 64) store.assign(elems + 64, 0);
    return (unsigned short *)(((uintptr_t)store.data() + 63) & ~(uintptr_t)63);
}

void tiles_to_bf16_t(const double *src, unsigned short *dst, long keys, long tiles, int cols, int cols_pad, long ldt, bool nt,
                     bool force_scalar = false) {
    constexpr int TK = 32, TG = 8;             // keys per tile; tiles per group: an image row gets TG x 64 contiguous bytes at a time
    const int ldtmp = (cols + 31) / 32 * 32;   // (one line per row and tile costs a DTLB miss and a DRAM page each: 4.9 -> x ms per 8192 keys)
    const bool simd = !force_scalar && have_avx512();
    // scratch of this thread: one group of rows, bf16, row-major (<= 256 x 1024 x 2 = 512 KiB), and one column block of its image
    unsigned short *tmp = vt_scratch((size_t)TG * TK * ldtmp + (size_t)32 * TG * TK + 64);
    unsigned short *obuf = tmp + (size_t)TG * TK * ldtmp;               // [32 columns][TG tiles][32 positions], 64-byte aligned
    for (long t0 = 0; t0 < tiles; t0 += TG) {
        const int tg = (int)(tiles - t0 < TG ? tiles - t0 : TG);
        const long r0 = t0 * TK, span = (long)tg * TK;
        const long nr = keys - r0 < span ? (keys - r0 > 0 ? keys - r0 : 0) : span;
        if (nr > 0) {
            if (simd) rows_to_bf16_avx512(src + r0 * cols, tmp, nr, cols, ldtmp, 1.0);
            else rows_to_bf16_scalar(src + r0 * cols, tmp, nr, cols, ldtmp, 1.0);
        }
        if (nr < span) memset(tmp + nr * ldtmp, 0, (size_t)(span - nr) * ldtmp * sizeof(unsigned short));
        unsigned short *out = dst + r0;                       // the group's first key position of image row 0
        const size_t run = (size_t)tg * TK;                   // elements of an image row this group fills
        for (int c0 = 0; c0 < cols; c0 += 32) {
            const int nc = cols - c0 < 32 ? cols - c0 : 32;
            for (int g = 0; g < tg; ++g) {                    // tile g of the group -> obuf[c][g][0..31]
                const unsigned short *in = tmp + (size_t)g * TK * ldtmp + c0;
                if (simd) {
                    transpose_block_32x32(in, ldtmp, obuf + (size_t)g * TK, (long)TG * TK, 32, false);
                } else {
                    for (int p = 0; p < TK; ++p) {
                        const unsigned short *row = in + (size_t)kvpos32(p) * ldtmp;     // position p holds key kvpos(p) (an involution)
                        for (int cc = 0; cc < 32; ++cc) obuf[((size_t)cc * TG + g) * TK + p] = row[cc];
                    }
                }
            }
            for (int cc = 0; cc < nc; ++cc) {
                unsigned short *line = out + (size_t)(c0 + cc) * ldt;
                const unsigned short *from = obuf + (size_t)cc * TG * TK;
#if defined(__x86_64__)
                if (nt && simd && ((uintptr_t)line & 63u) == 0) {
                    for (size_t i = 0; i < run; i += 32) stream_line_64(line + i, from + i);
                    continue;
                }
#endif
                memcpy(line, from, run * sizeof(unsigned short));
            }
        }
        for (int c = cols; c < cols_pad; ++c) memset(out + (size_t)c * ldt, 0, run * sizeof(unsigned short));
    }
#if defined(__x86_64__)
    if (nt && simd) _mm_sfence();
#endif
}

// The TILED Vt image (round 6, sdpa_internal.h; dv > 256): per 32-key tile one contiguous block [cols_pad / 512 chunks][512 columns]
// [32 key positions]; a column's 64-byte line holds the tile's keys in kvpos order, 16-byte chunk q at q ^ ((column >> 2) & 3).
// dst = the block of the first tile.  Per tile: 32 rows through the row converter into scratch, 32 x 32 blocks transposed in
// registers, each line's chunks permuted and streamed out -- 32 KiB of consecutive lines per (tile, chunk).
__attribute__((target("avx512f,avx512bw,avx512vl")))
inline void store_line_swz(unsigned short *line, const unsigned short *from, int x, bool nt) {
    __m512i y = _mm512_loadu_si512((const void *)from);
    switch (x) {                                     // chunk position p takes chunk p ^ x
        case 1: y = _mm512_shuffle_i32x4(y, y, 0xB1); break;
        case 2: y = _mm512_shuffle_i32x4(y, y, 0x4E); break;
        case 3: y = _mm512_shuffle_i32x4(y, y, 0x1B); break;
        default: break;
    }
    if (nt && ((uintptr_t)line & 63u) == 0) _mm512_stream_si512((__m512i *)line, y);
    else _mm512_storeu_si512((void *)line, y);
}

void tiles_to_bf16_tiled(const double *src, unsigned short *dst, long keys, long tiles, int cols, int cols_pad, bool nt,
                         bool force_scalar = false) {
    constexpr int TK = 32;
    const int ldtmp = (cols + 31) / 32 * 32;
    const bool simd = !force_scalar && have_avx512();
    unsigned short *tmp = vt_scratch((size_t)TK * ldtmp + (size_t)32 * TK + 64);
    unsigned short *obuf = tmp + (size_t)TK * ldtmp;                   // [32 columns][32 positions]
    const size_t tile_elems = (size_t)cols_pad * TK;
    for (long t = 0; t < tiles; ++t) {
        const long r0 = t * TK;
        const long nr = keys - r0 < TK ? (keys - r0 > 0 ? keys - r0 : 0) : TK;
        if (nr > 0) {
            if (simd) rows_to_bf16_avx512(src + r0 * cols, tmp, nr, cols, ldtmp, 1.0);
            else rows_to_bf16_scalar(src + r0 * cols, tmp, nr, cols, ldtmp, 1.0);
        }
        if (nr < TK) memset(tmp + nr * ldtmp, 0, (size_t)(TK - nr) * ldtmp * sizeof(unsigned short));
        unsigned short *tile = dst + (size_t)t * tile_elems;
        for (int c0 = 0; c0 < cols; c0 += 32) {
            const int nc = cols - c0 < 32 ? cols - c0 : 32;
            const unsigned short *in = tmp + c0;
            if (simd) {
                transpose_block_32x32(in, ldtmp, obuf, TK, 32, false);
                for (int cc = 0; cc < nc; ++cc) {
                    const int c = c0 + cc;
                    store_line_swz(tile + (size_t)(c >> 9) * (512 * TK) + (size_t)(c & 511) * TK, obuf + (size_t)cc * TK, (c >> 2) & 3, nt);
                }
            } else {
                for (int cc = 0; cc < nc; ++cc) {
                    const int c = c0 + cc, x = (c >> 2) & 3;
                    unsigned short *line = tile + (size_t)(c >> 9) * (512 * TK) + (size_t)(c & 511) * TK;
                    for (int p = 0; p < TK; ++p) line[(((p >> 3) ^ x) << 3) | (p & 7)] = in[(size_t)kvpos32(p) * ldtmp + cc];
                }
            }
        }
        for (int c = cols; c < cols_pad; ++c)
            memset(tile + (size_t)(c >> 9) * (512 * TK) + (size_t)(c & 511) * TK, 0, TK * sizeof(unsigned short));
    }
#if defined(__x86_64__)
    if (nt && simd) _mm_sfence();
#endif
}

// fp32 -> fp64 of `n` contiguous values: the reference's cvt_f2d_avx512 (attention-mpi.c:68-101, called on the
// root at :373 / :396), exact.  The AVX-512 form streams its stores past the cache once dst is 64-byte aligned:
// the caller reads `result` later, from another core as likely as not, and a write-allocate would first READ
// every line it is about to overwrite.
void widen_scalar(const float *src, double *dst, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = (double)src[i];
}

__attribute__((target("avx512f,avx512bw,avx512vl")))
void widen_avx512(const float *src, double *dst, size_t n) {
    size_t i = 0;
    while (i < n && ((uintptr_t)(dst + i) & 63u)) {
        dst[i] = (double)src[i];
        ++i;
    }
    for (; i + 8 <= n; i += 8) _mm512_stream_pd(dst + i, _mm512_cvtps_pd(_mm256_loadu_ps(src + i)));
    for (; i < n; ++i) dst[i] = (double)src[i];
    _mm_sfence();
}

inline void widen_range(const float *src, double *dst, size_t n, bool force_scalar = false) {
    if (!force_scalar && have_avx512()) widen_avx512(src, dst, n);
    else widen_scalar(src, dst, n);
}

// one blocking widen() call: the calling thread and every worker that wakes up in time take items from it
struct WidenJob {
    const float *src = nullptr;
    double *dst = nullptr;
    size_t n = 0, per = 0, items = 0;
    std::atomic<size_t> next{0};
    std::atomic<size_t> left{0};
    void work() {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= items) return;
            const size_t e0 = i * per, cnt = n - e0 < per ? n - e0 : per;
            widen_range(src + e0, dst + e0, cnt);
            left.fetch_sub(1, std::memory_order_release);
        }
    }
};

struct Task {
    const double *src;
    void *dst;
    int cols, ld;
    CvtKind kind;
    double mult;
    std::atomic<long> remaining;
    // kCvtBf16T only: rows of the entry (the image holds whole 32-key tiles up to the padded count), image rows, row stride
    long keys = 0, ldt = 0;
    int cols_pad = 0;
    bool tiled = false;      // kCvtBf16T: the tiled Vt image (cols > 256)
    int swz = 0;             // kCvtBf16Swz: the K image's chunk swizzle mask
    Task(const double *s, void *d, int c, int l, CvtKind k, double mu, long items)
        : src(s), dst(d), cols(c), ld(l), kind(k), mult(mu), remaining(items) {}
};

struct Item {
    int task;
    long row0, rows;
};

// everything one call converts: the worker threads take a reference when they wake up, so a thread that
// wakes late for a finished call finds an exhausted work list instead of the next call's half-built one
struct Batch {
    std::deque<Task> tasks;            // (deque: a Task holds an atomic and never moves)
    std::vector<Item> items;
    std::atomic<long> next{0};
    // $SDPA_DEBUG host_cvt_trace only (microseconds on the steady clock)
    double t_kick = 0;
    std::atomic<double> first_min{1e300}, first_max{0}, last_done{0}, busy_us{0};
    std::atomic<int> workers{0};
};

inline double trace_now() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline void atomic_min(std::atomic<double> &a, double v) { double c = a.load(); while (v < c && !a.compare_exchange_weak(c, v)) {} }
inline void atomic_max(std::atomic<double> &a, double v) { double c = a.load(); while (v > c && !a.compare_exchange_weak(c, v)) {} }
inline void atomic_add(std::atomic<double> &a, double v) { double c = a.load(); while (!a.compare_exchange_weak(c, c + v)) {} }

// ---- where the pool's threads run ($SDPA_DEBUG host_cvt_pin=1, OPT-IN: a measured negative result) -----------------------------------
// On the GPU box's two-socket host a standalone probe reads the fp64 source at 250-420 GB/s from threads pinned one per core on
// the source pages' NUMA node and at 110 GB/s flat from the other one (tools/probes/hostcvt_placement_probe.cpp), and one-shot CLI
// runs of config 5 in bf16 are bimodal box to box (6.6-7.4 ms against 9.3-9.5).  Confining the pool to the source's node -- a few
// pages of the call's arrays are asked for their node (move_pages with no target: a query), the threads get that node's CPUs, or
// all CPUs again when the samples disagree -- does NOT fix that: warm calls are unchanged (6.42-6.60 against 6.46-6.68 ms), and
// three of five cold CLI runs took 20-28 ms with the pool confined (kv stage 13-22 ms) where every unconfined one took 6.6-7.1
// (profiles/r05/converter_pool_numa_pin_ab.log; the box's container has a CPU quota of 16 cores -- 32 runnable threads on one
// socket's CPUs meet it differently than threads spread over both).  Kept behind the knob, off by default.
struct NumaMap {
    std::vector<cpu_set_t> node_cpus;      // per node: its CPUs that this process may use
    cpu_set_t allowed;
    bool usable = false;
    NumaMap() {
        CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
        for (int nd = 0; nd < 64; ++nd) {
            char path[128];
            snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", nd);
            FILE *f = fopen(path, "r");
            if (!f) break;
            char buf[4096];
            cpu_set_t set;
            CPU_ZERO(&set);
            if (fgets(buf, sizeof buf, f)) {
                char *p = buf;
                while (*p && *p != '\n') {
                    long a = strtol(p, &p, 10), b = a;
                    if (*p == '-') b = strtol(p + 1, &p, 10);
                    for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
                        if (c >= 0 && CPU_ISSET((int)c, &allowed)) CPU_SET((int)c, &set);
                    if (*p == ',') ++p;
                    else if (*p != '\n' && *p != 0 && *p != '-') break;
                }
            }
            fclose(f);
            node_cpus.push_back(set);
        }
        usable = node_cpus.size() > 1;
    }
    // node of the page that holds p, or -1
    static int node_of(const void *p) {
        void *page = (void *)((uintptr_t)p & ~(uintptr_t)4095);
        int status = -1;
        const long rc = syscall(SYS_move_pages, 0, 1UL, &page, nullptr, &status, 0);
        return rc == 0 ? status : -1;
    }
};

class Pool final : public HostConverter {
public:
    ~Pool() override {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
        for (Buf &b : buf_)
            if (b.p && hipHostFree(b.p) != hipSuccess) (void)hipGetLastError();
    }
    // confine the workers to the NUMA node the call's source pages live on (class comment above); a handful of syscalls per call
    void place_near(const void *const *arrays, const size_t *bytes, int n_arrays) override {
        if (!pin_ || !numa_.usable || th_.empty()) return;
        int node = -2;                                      // -2: nothing sampled yet; -1: unknown or mixed
        for (int a = 0; a < n_arrays && node != -1; ++a) {
            if (!arrays[a] || bytes[a] == 0) continue;
            for (int k = 0; k < 4 && node != -1; ++k) {
                const char *p = (const char *)arrays[a] + (k == 3 ? bytes[a] - 1 : (bytes[a] - 1) / 3 * (size_t)k);    // first, 1/3, 2/3, last byte
                const int nd = NumaMap::node_of(p);
                if (nd < 0 || nd >= (int)numa_.node_cpus.size()) node = -1;
                else if (node == -2) node = nd;
                else if (node != nd) node = -1;
            }
        }
        if (node == -2) node = -1;
        if (node >= 0 && CPU_COUNT(&numa_.node_cpus[node]) < 4) node = -1;       // (a node this process may hardly use)
        if (node == placed_) return;
        const cpu_set_t &set = node >= 0 ? numa_.node_cpus[node] : numa_.allowed;
        for (std::thread &t : th_) (void)pthread_setaffinity_np(t.native_handle(), sizeof set, &set);
        placed_ = node;
    }
    int placed_node() const override { return placed_; }
    bool start(int n) {
        for (int i = 0; i < n; ++i) th_.emplace_back([this] { loop(); });
        return true;
    }
    int threads() const override { return (int)th_.size(); }

    void *staging(int which, size_t bytes) override {
        if (which < 0 || which > 3) return nullptr;
        Buf &b = buf_[which];
        if (b.cap >= bytes && b.p) return b.p;
        if (b.p && hipHostFree(b.p) != hipSuccess) (void)hipGetLastError();
        b.p = nullptr;
        b.cap = 0;
        const size_t want = bytes + bytes / 8 + 4096;      // some slack: the next problem is often a bit larger
        if (hipHostMalloc(&b.p, want, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            b.p = nullptr;
            return nullptr;
        }
        b.cap = want;
        return b.p;
    }

    void begin() override {
        finish();
        mine_ = std::make_shared<Batch>();
    }
    int submit(const double *src, void *dst, long rows, int cols, int ld, CvtKind kind, double mult) override {
        if (!mine_) mine_ = std::make_shared<Batch>();
        // items of ~64 KiB of source ($SDPA_DEBUG host_cvt_item_kb): small enough to balance, large enough to stream
        long per = (long)item_kb_ * 128 / (cols > 0 ? cols : 1);
        if (per < 1) per = 1;
        const long n_items = rows > 0 ? (rows + per - 1) / per : 0;
        const int id = (int)mine_->tasks.size();
        mine_->tasks.emplace_back(src, dst, cols, ld, kind, mult, n_items);
        if (kind == kCvtBf16Swz) mine_->tasks.back().swz = ld / 8 >= 16 ? 15 : ld / 8 - 1;
        for (long r = 0; r < rows; r += per) mine_->items.push_back({id, r, rows - r < per ? rows - r : per});
        return id;
    }
    int submit_t(const double *src, unsigned short *dst, long keys, long keys_pad, int cols, int cols_pad, long ldt) override {
        if (!mine_) mine_ = std::make_shared<Batch>();
        // items of whole 32-key tiles, ~$SDPA_DEBUG host_cvt_item_kb of source each (at least one tile)
        long per = (long)item_kb_ * 128 / (cols > 0 ? cols : 1);
        per = per < 32 ? 32 : (per + 31) / 32 * 32;
        const long total = (keys_pad + 31) / 32 * 32;
        const long n_items = total > 0 ? (total + per - 1) / per : 0;
        const int id = (int)mine_->tasks.size();
        mine_->tasks.emplace_back(src, (void *)dst, cols, 0, kCvtBf16T, 1.0, n_items);
        Task &t = mine_->tasks.back();
        t.keys = keys; t.ldt = ldt; t.cols_pad = cols_pad; t.tiled = cols > 256;
        for (long r = 0; r < total; r += per) mine_->items.push_back({id, r, total - r < per ? total - r : per});
        return id;
    }
    void kick() override {
        if (trace_ && mine_) mine_->t_kick = trace_now();
        {
            std::lock_guard<std::mutex> lk(mu_);
            cur_ = mine_;               // complete: nothing is added to a batch after this
            ++gen_;
            gen_pub_.store(gen_, std::memory_order_release);
        }
        cv_.notify_all();
    }
    void wait(int task) override {
        if (!mine_ || task < 0 || task >= (int)mine_->tasks.size()) return;
        while (mine_->tasks[task].remaining.load(std::memory_order_acquire) > 0) relax();
    }
    // every item of the batch has been converted: no thread reads the caller's arrays any more
    void finish() override {
        if (!mine_) return;
        bool kicked;
        {
            std::lock_guard<std::mutex> lk(mu_);
            kicked = cur_ == mine_;
        }
        if (kicked)
            for (Task &t : mine_->tasks)
                while (t.remaining.load(std::memory_order_acquire) > 0) relax();
        if (trace_ && kicked && !mine_->items.empty()) {
            // $SDPA_DEBUG host_cvt_trace=1: where one call's conversions spent their time (one line on stderr per call)
            Batch &b = *mine_;
            double bytes = 0;
            for (const Item &it : b.items) bytes += (double)it.rows * b.tasks[it.task].cols * 8.0;
            const double first_lo = b.first_min.load(), first_hi = b.first_max.load(), last = b.last_done.load();
            fprintf(stderr, "sdpa hostcvt trace: %zu items, %.1f MB of fp64 | workers that took items %d of %zu | first item taken %.0f .. %.0f us "
                    "after the kick | last item done at %.0f us | sum of busy time %.0f us (%.1f GB/s per busy thread) | %.1f GB/s over the span | workers on NUMA node %d\n",
                    b.items.size(), bytes / 1e6, b.workers.load(), th_.size(), first_lo - b.t_kick, first_hi - b.t_kick, last - b.t_kick,
                    b.busy_us.load(), bytes / 1e3 / std::max(1.0, b.busy_us.load()), bytes / 1e3 / std::max(1.0, last - b.t_kick), placed_);
        }
        mine_.reset();
    }

    // blocking: dst[i] = (double)src[i], i < n, spread over the pool; the caller works too, so a pool whose threads
    // are slow to wake costs nothing but their share
    void widen(const float *src, double *dst, size_t n) override {
        if (n == 0) return;
        const size_t per = 16384;                        // 64 KiB read, 128 KiB written per item
        const size_t items = (n + per - 1) / per;
        if (items < 3 || th_.empty()) {
            widen_range(src, dst, n);
            return;
        }
        auto j = std::make_shared<WidenJob>();
        j->src = src; j->dst = dst; j->n = n; j->per = per; j->items = items;
        j->left.store(items, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu_);
            wjob_ = j;
            ++gen_;
            gen_pub_.store(gen_, std::memory_order_release);
        }
        cv_.notify_all();
        j->work();
        while (j->left.load(std::memory_order_acquire) > 0) relax();
        std::lock_guard<std::mutex> lk(mu_);
        if (wjob_ == j) wjob_.reset();
    }

private:
    struct Buf {
        void *p = nullptr;
        size_t cap = 0;
    };
    void loop() {
        (void)vt_scratch((size_t)8 * 32 * 512 + (size_t)32 * 8 * 32 + 64);       // (the transposing converter's scratch at dv = 512, touched now)
        unsigned long seen = 0;
        bool hot = false;
        for (;;) {
            std::shared_ptr<Batch> b;
            std::shared_ptr<WidenJob> w;
            // the widen jobs of a call come in quick succession (one per piece of result rows): behind one, look
            // for the next for a moment before going to sleep
            for (int spin = hot ? 20000 : 0; spin > 0 && gen_pub_.load(std::memory_order_acquire) == seen; --spin) relax();
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                b = cur_;
                w = wjob_;
            }
            hot = (bool)w;
            if (w) w->work();
            if (!b) continue;
            const long total = (long)b->items.size();
            double t_first = 0;
            for (;;) {
                const long i = b->next.fetch_add(1, std::memory_order_relaxed);
                if (i >= total) break;
                if (trace_ && t_first == 0) t_first = trace_now();
                const Item &it = b->items[i];
                Task &t = b->tasks[it.task];
                const double *s = t.src + it.row0 * t.cols;
                if (t.kind == kCvtF32)
                    rows_to_f32(s, (float *)t.dst + it.row0 * t.ld, it.rows, t.cols, t.ld, nt_);
                else if (t.kind == kCvtBf16T && t.tiled)
                    tiles_to_bf16_tiled(s, (unsigned short *)t.dst + (size_t)it.row0 * t.cols_pad, t.keys - it.row0, it.rows / 32, t.cols, t.cols_pad, nt_);
                else if (t.kind == kCvtBf16T)          // it.row0, it.rows: whole tiles; keys left of the entry from row0 on
                    tiles_to_bf16_t(s, (unsigned short *)t.dst + it.row0, t.keys - it.row0, it.rows / 32, t.cols, t.cols_pad, t.ldt, nt_);
                else if (t.kind == kCvtBf16Swz)        // (the task's dst is an image row that is a multiple of 16)
                    rows_to_bf16_swz(s, (unsigned short *)t.dst + it.row0 * t.ld, it.rows, t.cols, t.ld, t.mult, it.row0, t.swz, nt_);
                else
                    rows_to_bf16(s, (unsigned short *)t.dst + it.row0 * t.ld, it.rows, t.cols, t.ld, t.mult, nt_);
                t.remaining.fetch_sub(1, std::memory_order_release);
            }
            if (trace_ && t_first != 0) {
                const double t_end = trace_now();
                atomic_min(b->first_min, t_first);
                atomic_max(b->first_max, t_first);
                atomic_max(b->last_done, t_end);
                atomic_add(b->busy_us, t_end - t_first);
                b->workers.fetch_add(1);
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
    unsigned long gen_ = 0;
    std::atomic<unsigned long> gen_pub_{0};   // gen_, readable without the lock (the workers' short spin)
    std::shared_ptr<WidenJob> wjob_;   // the widen() call in flight, if any (guarded by mu_)
    std::shared_ptr<Batch> cur_;       // what the threads work on (guarded by mu_)
    std::shared_ptr<Batch> mine_;      // the calling thread's handle on the batch it is building / waiting for
    Buf buf_[4];
    const bool nt_ = stream_stores_default();
    const bool trace_ = sdpa_debug_int("host_cvt_trace", 0) != 0;
    const bool pin_ = sdpa_debug_int("host_cvt_pin", 0) != 0;       // opt-in: see the class comment
    NumaMap numa_;
    int placed_ = -1;                  // the node the workers are confined to (-1: every allowed CPU)
    const int item_kb_ = [] {
        const int kb = sdpa_debug_int("host_cvt_item_kb", 64);
        return kb < 4 ? 4 : kb > 16384 ? 16384 : kb;
    }();
};

}  // namespace

// one thread, `rows` rows: the conversion the pool's threads run (also the C ABI's sdpa_host_cvt_rows)
void host_convert_rows(const double *src, void *dst, long rows, int cols, int ld, CvtKind kind, double mult,
                       bool force_scalar, int stream_stores) {
    const bool nt = stream_stores < 0 ? stream_stores_default() : stream_stores != 0;
    if (kind == kCvtF32) {
        if (force_scalar) rows_to_f32_scalar(src, (float *)dst, rows, cols, ld);
        else rows_to_f32(src, (float *)dst, rows, cols, ld, nt);
    } else if (kind == kCvtBf16Swz) {
        rows_to_bf16_swz(src, (unsigned short *)dst, rows, cols, ld, mult, 0, ld / 8 >= 16 ? 15 : ld / 8 - 1, nt && !force_scalar, force_scalar);
    } else {
        if (force_scalar) rows_to_bf16_scalar(src, (unsigned short *)dst, rows, cols, ld, mult);
        else rows_to_bf16(src, (unsigned short *)dst, rows, cols, ld, mult, nt);
    }
}

void host_convert_vt(const double *src, unsigned short *dst, long keys, long keys_pad, int cols, int cols_pad, long ldt,
                     bool force_scalar, int stream_stores) {
    const bool nt = stream_stores < 0 ? stream_stores_default() : stream_stores != 0;
    if (cols > 256) tiles_to_bf16_tiled(src, dst, keys, (keys_pad + 31) / 32, cols, cols_pad, nt && !force_scalar, force_scalar);
    else tiles_to_bf16_t(src, dst, keys, (keys_pad + 31) / 32, cols, cols_pad, ldt, nt && !force_scalar, force_scalar);
}

void host_widen(const float *src, double *dst, size_t n, bool force_scalar) { widen_range(src, dst, n, force_scalar); }

HostConverter *HostConverter::create(int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    Pool *p = new Pool;
    if (!p->start(threads)) {
        delete p;
        return nullptr;
    }
    return p;
}

}  // namespace sdpa
