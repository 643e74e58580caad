// sdpa_fwd_f32_dksplit.hip -- the fp32 fused kernel for head dims beyond one wave's register file:
// 256 < dk <= 1024 (and non-dense 128 < dk <= 256 with dv > 128).  Same stage, same outputs as the
// kernels of sdpa_fwd_f32.hip (online_softmax_attention, attention-mpi.c:168-189); its own translation
// unit so that work on it does not touch the headline kernel's object code.
#include "sdpa_f32_device.h"

#include <math.h>
#include <atomic>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

SDPA_AUDIT_COUNTER(g_dks_audit)

namespace sdpa {

// ---------------------------------------------------------------------------
// 256 < dk <= 512 in fp32 (and 128 < dk <= 256 when dv > 128, where it saves the per-chunk score
// recompute).  A 32 x 512 fp32 Q fragment alone is 256 registers, so here the four waves of a
// workgroup SPLIT the contraction dimensions between them instead of the query rows:
//   * workgroup = QB 32-row MFMA blocks of query rows: two (64 rows) for dk <= 512, one for 512 < dk <= 1024,
//     where the Q slice of ONE block already is 96 / 128 registers.  Wave w holds Q[:, DKS*w .. DKS*(w+1))
//     (128 registers at DKS = 128) and the O^T slice of V columns [DVS*w, DVS*(w+1)) of the chunk (<= 128 regs);
//   * per 32-key tile a wave computes the PARTIAL score tile over its dk slice -- K fragments come
//     straight from global memory, nobody else needs them -- the four partials go through LDS and
//     every wave sums them in the same fixed order (bitwise the same S^T in all four, so the
//     replicated online-softmax state stays in step), then each wave accumulates its dv slice,
//     V fragments straight from global memory as well;
//   * no K/V tile in LDS at all; one barrier per tile (the exchange buffer is double-buffered).
// 256 MFMAs of 64 cycles per wave and tile against 32 KiB of global reads: matrix-pipe bound.
// Same outputs and the same online-softmax arithmetic as fused_partial_kernel.
// ---------------------------------------------------------------------------
// (Rounds 1-5 also carried the serial-phase form of this kernel, fused_dksplit_kernel -- partial S, exchange, replicated softmax,
// P.V one after the other, ~75 % of the pipelined form's rate -- as its bit-identity twin behind $SDPA_DKSPLIT_PIPE=0.  Retired in
// round 6: no shape takes it; the pipelined form below is checked against the fp64 oracle and itself.)
// compile-time loop: f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>).  The pipelined kernel's
// units are selected by `if constexpr` on the index, so nothing depends on the unroller's size thresholds.
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ---------------------------------------------------------------------------
// Software-pipelined across tiles (round 3).  Run one after the other in every wave -- partial S, exchange, replicated
// softmax, P.V -- the phases leave the matrix pipe idle while a wave sums the exchanged partials and exponentiates (ONE wave
// per SIMD: the Q slice and the O^T slice fill the register file): ~75 % at dk = dv = 512.  Here the exchange sums and
// the softmax of tile t+1 are cut into units of a few instructions and placed BETWEEN the P.V MFMAs of
// tile t, in program order (a wave stalls at an MFMA issue while the pipe is busy, so only instructions
// written between two MFMAs run in the first one's shadow):
//   A(t):  partial S(t+1) over this wave's dk slice -> own slot of the exchange buffer; barrier
//   B(t):  O^T += V(t)^T P(t)^T, and between its MFMAs: read the four partials of S(t+1), sum them in
//          the fixed order, row max, alpha, exponentials -> P(t+1), relative to the new running max
//   then:  O *= alpha(t+1) where a maximum rose (after P.V(t) has landed)
// The ragged last tile (masked scores) and the first tile take the units without MFMAs in between.
// ---------------------------------------------------------------------------
template <int DKS, int DVS, int QB>
__global__ __launch_bounds__(256, 1) void fused_dksplit_pipe_kernel(
    PartialArgs a, int kv_per_split, int n_qblocks, int n_chunks, float scale) {
    SDPA_AUDIT_LAUNCH(g_dks_audit);
    static_assert(QB == 1 || QB == 2, "one or two query blocks");
    constexpr int ROWS = 32 * QB;
    constexpr int NU = DKS / 8;
    constexpr int NT = DVS / 32;
    constexpr int XLD = 20;
    constexpr int XBUF = 4 * QB * 64 * XLD;
    constexpr int PD = 4;
    constexpr int G = 4 * QB;                  // exchange groups: 4 score registers of one query block each
    constexpr int MPS = NT * QB;               // MFMAs per P.V k-step
    [[maybe_unused]] constexpr int SLOTS = 16 * MPS;            // MFMAs of one tile's P.V = places to put a unit
    constexpr int U_X = 7 * G;                 // units: per group 4 reads (one per wave's partial) + 3 sums
    constexpr int U_M = 6 * QB;                // row max (5 units of 3 max) + running-max update, per block
    constexpr int U_E = 16 * QB;               // one exponential each
    constexpr int UNITS = U_X + U_M + U_E;
    constexpr int USLOTS = 14 * MPS;           // step 0 carries the exchange stores, the barrier sits behind step 1
    constexpr int UPS = (UNITS + USLOTS - 1) / USLOTS;
    constexpr int WPS = (4 * QB + MPS - 1) / MPS;      // exchange stores per MFMA of step 0

    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][XBUF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int hi = lane >> 5;

    int work = xcd_remap(blockIdx.x, gridDim.x);
    const int qblock = work % n_qblocks;
    work /= n_qblocks;
    const int chunk = work % n_chunks;
    const int split = work / n_chunks;
    const int dk0 = wave * DKS;
    const int dvw0 = chunk * (4 * DVS) + wave * DVS;

    const int kv_begin = split * kv_per_split;
    const int kv_end = min(a.n_local, kv_begin + kv_per_split);
    const int ntiles = kv_end > kv_begin ? (kv_end - kv_begin + kKvTile - 1) / kKvTile : 0;
    const int nfull = kv_end > kv_begin ? (kv_end - kv_begin) / kKvTile : 0;
    const float c = scale * 1.44269504088896340736f;

    f32x4 qf[QB][NU];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = qblock * ROWS + qb * 32 + li;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int col = dk0 + 8 * u + 4 * hi;
            qf[qb][u] = (qrow < a.m && col < a.ldq)
                            ? *reinterpret_cast<const f32x4 *>(a.Q + (size_t)qrow * a.ldq + col)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
        if constexpr (QB == 2) asm volatile("" : "+a"(qf[QB - 1][u]));

    f32x16 oacc[NT][QB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][qb][r] = 0.f;
    auto pin_o = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            asm volatile("" : "+a"(oacc[tt][0]));
            if constexpr (QB == 2) asm volatile("" : "+a"(oacc[tt][QB - 1]));
        }
    };
    pin_o();

    const unsigned kcol0 = (unsigned)(dk0 + 4 * hi) * 4u, kcol_last = (unsigned)(a.ldk - 4) * 4u;
    auto kcolb = [&](int u) __attribute__((always_inline)) -> unsigned { return min(kcol0 + 32u * u, kcol_last); };
    const unsigned vcolb = (unsigned)min(dvw0 + NT * li, a.ldv - NT) * 4u;

    // P(t) of the tile whose P.V runs (s*), and the tile in the making (n*: partial S, S, then P)
    f32x16 s0, s1, n0, n1;
    float m_run0 = -INFINITY, m_run1 = -INFINITY, l_run0 = 0.f, l_run1 = 0.f;
    float tmax0 = 0.f, tmax1 = 0.f, alpha0 = 1.f, alpha1 = 1.f, mc0 = 0.f, mc1 = 0.f;
    bool rise = false;
    f32x4 rd[4];
    VRun<NT> vq[PD];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; n0[r] = 0.f; n1[r] = 0.f; }

    // first K fragments of a tile (the ring partial_scores continues).  Issued a whole phase ahead -- under the
    // previous tile's P.V -- like the first V fragments under the score MFMAs: a load used right after its issue
    // costs its full latency once per tile, and with one wave per SIMD nobody covers it.
    f32x4 kq[PD];
    auto first_k = [&](int tile) __attribute__((always_inline)) {
        const int base = kv_begin + tile * kKvTile;
        const int last = kv_end - 1 - base;
        const char *kb = reinterpret_cast<const char *>(a.K + (size_t)base * a.ldk);
        const unsigned krow = (unsigned)min(li, last) * (unsigned)a.ldk * 4u;
#pragma unroll
        for (int i = 0; i < PD; ++i) kq[i] = *reinterpret_cast<const f32x4 *>(SDPA_AUDITED_PTR(g_dks_audit, kb + (krow + kcolb(i)), 16, a.K, a.K + (size_t)a.n_local * a.ldk));
    };
    // ---- A: partial S^T of tile `tile` over this wave's dk slice -> n0 / n1 (first_k(tile) went out a phase ago)
    auto partial_scores = [&](int tile) __attribute__((always_inline)) {
        const int base = kv_begin + tile * kKvTile;
        const char *kb = reinterpret_cast<const char *>(a.K + (size_t)base * a.ldk);
        const unsigned krow = (unsigned)min(li, kv_end - 1 - base) * (unsigned)a.ldk * 4u;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const f32x4 kf = kq[u % PD];
            if (u + PD < NU) kq[u % PD] = *reinterpret_cast<const f32x4 *>(SDPA_AUDITED_PTR(g_dks_audit, kb + (krow + kcolb(u + PD)), 16, a.K, a.K + (size_t)a.n_local * a.ldk));
            __builtin_amdgcn_sched_barrier(0);
            n0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[0][u].x, u == 0 ? zero : n0, 0, 0, 0);
            if constexpr (QB == 2) n1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[QB - 1][u].x, u == 0 ? zero : n1, 0, 0, 0);
            n0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[0][u].y, n0, 0, 0, 0);
            if constexpr (QB == 2) n1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[QB - 1][u].y, n1, 0, 0, 0);
            n0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[0][u].z, n0, 0, 0, 0);
            if constexpr (QB == 2) n1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[QB - 1][u].z, n1, 0, 0, 0);
            n0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[0][u].w, n0, 0, 0, 0);
            if constexpr (QB == 2) n1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[QB - 1][u].w, n1, 0, 0, 0);
        }
    };
    // piece i (of 4 * QB) of this wave's partial -> its slot of tile's exchange buffer (block 0's four first)
    auto store_partial = [&](auto I, int tile) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, qb = i / 4, q4 = i % 4;
        float *mine = smem + (tile & 1) * XBUF + ((wave * QB + qb) * 64 + lane) * XLD;
        if constexpr (qb == 0)
            *reinterpret_cast<f32x4 *>(mine + 4 * q4) = f32x4{n0[4 * q4], n0[4 * q4 + 1], n0[4 * q4 + 2], n0[4 * q4 + 3]};
        else
            *reinterpret_cast<f32x4 *>(mine + 4 * q4) = f32x4{n1[4 * q4], n1[4 * q4 + 1], n1[4 * q4 + 2], n1[4 * q4 + 3]};
    };
    auto store_partials = [&](int tile) __attribute__((always_inline)) {
        static_for<0, 4 * QB>([&](auto I) __attribute__((always_inline)) { store_partial(I, tile); });
    };

    // first V fragments of a tile (the ring the P.V loop continues)
    auto first_v = [&](int tile) __attribute__((always_inline)) {
        const int base = kv_begin + tile * kKvTile;
        const int last = kv_end - 1 - base;
        const char *vb = reinterpret_cast<const char *>(a.V + (size_t)base * a.ldv);
#pragma unroll
        for (int i = 0; i < PD; ++i)
            vq[i] = VRun<NT>::load(reinterpret_cast<const float *>(
                SDPA_AUDITED_PTR(g_dks_audit, vb + ((unsigned)min(crow(i, 0) + 4 * hi, last) * (unsigned)a.ldv * 4u + vcolb), NT * 4, a.V, a.V + (size_t)a.n_local * a.ldv)));
    };

    // ---- the exchange sums and the online softmax of the tile in n0 / n1, in UNITS pieces (k in order)
    auto unit = [&](auto K, const float *xb) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        if constexpr (k < U_X) {
            constexpr int j = k / 7, x = k % 7, q4 = j / QB, qb = j % QB;
            if constexpr (x < 4) {           // wave x's partial of group j
                rd[x] = *reinterpret_cast<const f32x4 *>(xb + ((x * QB + qb) * 64 + lane) * XLD + 4 * q4);
            } else {                         // ((p0 + p1) + p2) + p3, as the serial kernel
                constexpr int w = x - 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (qb == 0)
                        n0[4 * q4 + e] = (w == 1) ? pinned_add(rd[0][e], rd[1][e]) : pinned_add(n0[4 * q4 + e], rd[w][e]);
                    else
                        n1[4 * q4 + e] = (w == 1) ? pinned_add(rd[0][e], rd[1][e]) : pinned_add(n1[4 * q4 + e], rd[w][e]);
                }
            }
        } else if constexpr (k < U_X + U_M) {
            // (max, exp2 and the row-sum adds as volatile asm: pure VALU work otherwise leaves the MFMA shadow
            //  it was written in -- LLVM sinks it to the loop end, next to its first use)
            constexpr int qb = (k - U_X) / 6, x = (k - U_X) % 6;
            f32x16 &n = qb == 0 ? n0 : n1;
            float &tm = qb == 0 ? tmax0 : tmax1;
            if constexpr (x < 5) {
                tm = pinned_max3(x == 0 ? n[0] : tm, n[3 * x + 1], n[3 * x + 2]);
                tm = pinned_max(tm, n[3 * x + 3]);
            } else {
                float &m_run = qb == 0 ? m_run0 : m_run1;
                float &l_run = qb == 0 ? l_run0 : l_run1;
                float &alpha = qb == 0 ? alpha0 : alpha1;
                float &mc = qb == 0 ? mc0 : mc1;
                tm = fmaxf(tm, __shfl_xor(tm, 32));
                const float m_new = fmaxf(m_run, tm);
                const bool up = m_new > m_run;
                alpha = up ? pinned_exp2((m_run - m_new) * c) : 1.f;
                rise = qb == 0 ? up : (rise || up);
                l_run *= alpha;
                m_run = m_new;
                mc = m_run * c;
            }
        } else {
            constexpr int qb = (k - U_X - U_M) / 16, r = (k - U_X - U_M) % 16;
            if constexpr (qb == 0) {
                n0[r] = pinned_exp2(fmaf(n0[r], c, -mc0));
                l_run0 = pinned_add(l_run0, n0[r]);
            } else {
                n1[r] = pinned_exp2(fmaf(n1[r], c, -mc1));
                l_run1 = pinned_add(l_run1, n1[r]);
            }
        }
    };
    // all units back to back (first tile, ragged last tile); `valid` < 32 masks the key rows past the shard end
    auto units_serial = [&](int tile, int valid) __attribute__((always_inline)) {
        const float *xb = smem + (tile & 1) * XBUF;
        static_for<0, U_X>([&](auto K) __attribute__((always_inline)) { unit(K, xb); });
        if (valid < kKvTile) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (crow(r, hi) >= valid) {
                    n0[r] = -INFINITY;
                    if constexpr (QB == 2) n1[r] = -INFINITY;
                }
        }
        static_for<U_X, UNITS>([&](auto K) __attribute__((always_inline)) { unit(K, xb); });
    };

    // ---- B: O^T slice += V(tile)^T . P(tile)^T.  With NEXT (a full tile + 1 follows, its partial scores in n0 / n1):
    // step 0 carries the exchange stores of tile + 1, the barrier sits behind step 1, the units of tile + 1 follow
    // between the MFMAs of steps 2..15.
    auto pv = [&](auto next, int tile) __attribute__((always_inline)) {
        constexpr bool NEXT = decltype(next)::value;
        const int base = kv_begin + tile * kKvTile;
        const int last = kv_end - 1 - base;
        const char *vb = reinterpret_cast<const char *>(a.V + (size_t)base * a.ldv);
        const float *xb = smem + ((tile + 1) & 1) * XBUF;
        static_for<0, 16>([&](auto R) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
            const VRun<NT> vf = vq[r % PD];
            if constexpr (r + PD < 16)
                vq[r % PD] = VRun<NT>::load(reinterpret_cast<const float *>(
                SDPA_AUDITED_PTR(g_dks_audit, vb + ((unsigned)min(crow(r + PD, 0) + 4 * hi, last) * (unsigned)a.ldv * 4u + vcolb), NT * 4, a.V, a.V + (size_t)a.n_local * a.ldv)));
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, MPS>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value, tt = i / QB, qb = i % QB;
                if constexpr (qb == 0)
                    oacc[tt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.v[tt], s0[r], oacc[tt][0], 0, 0, 0);
                else
                    oacc[tt][QB - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.v[tt], s1[r], oacc[tt][QB - 1], 0, 0, 0);
                if constexpr (NEXT) {
                    if constexpr (r == 0) {
                        static_for<i * WPS, (i + 1) * WPS < 4 * QB ? (i + 1) * WPS : 4 * QB>(
                            [&](auto W) __attribute__((always_inline)) { store_partial(W, tile + 1); });
                    } else if constexpr (r == 1) {
                        if constexpr (i == MPS - 1) __syncthreads();
                    } else {
                        constexpr int slot = (r - 2) * MPS + i;
                        static_for<slot * UPS, (slot + 1) * UPS < UNITS ? (slot + 1) * UPS : UNITS>(
                            [&](auto K) __attribute__((always_inline)) { unit(K, xb); });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
    };
    // the running maxima that rose while tile's P was made: rescale O before that tile's P.V
    auto rescale = [&]() __attribute__((always_inline)) {
        if (__any(rise)) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    oacc[tt][0][r] *= alpha0;
                    if constexpr (QB == 2) oacc[tt][QB - 1][r] *= alpha1;
                }
        }
    };
    auto adopt = [&]() __attribute__((always_inline)) {
        s0 = n0;
        if constexpr (QB == 2) s1 = n1;
    };


    if (ntiles > 0) {
        // tile 0: nothing to overlap with yet (O is zero: no rescale)
        first_k(0);
        partial_scores(0);
        store_partials(0);
        __syncthreads();
        if (ntiles > 1) first_k(1);
        units_serial(0, kv_end - kv_begin);
        adopt();
        pin_o();
        int cur = 0;
        for (; cur + 1 < nfull; ++cur) {       // the next tile is a full one: its softmax rides under this tile's P.V
            pin_o();
            first_v(cur);                      // in flight under the score MFMAs
            partial_scores(cur + 1);
            pin_o();
            first_k(min(cur + 2, ntiles - 1)); // in flight under the P.V MFMAs (the last one is a harmless re-read)
            pv(std::true_type{}, cur);
            pin_o();
            rescale();
            adopt();
            pin_o();
        }
        if (cur + 1 < ntiles) {                // a ragged last tile: masked, taken without the overlap
            pin_o();
            first_v(cur);
            partial_scores(cur + 1);
            store_partials(cur + 1);
            pin_o();
            __syncthreads();
            pv(std::false_type{}, cur);
            pin_o();
            units_serial(cur + 1, kv_end - (kv_begin + (cur + 1) * kKvTile));
            rescale();
            adopt();
            pin_o();
            ++cur;
        }
        first_v(cur);
        pv(std::false_type{}, cur);
        pin_o();
    }

    // ---- epilogue
    float *out = a.contrib;
    float *omax = a.lmax, *osum = a.lsum;
    int ldo = a.ldo;
    if (a.kv_splits > 1) {
        ldo = a.ws_ld;
        out = a.ws_contrib + (size_t)split * a.ws_rows * ldo;
        omax = a.ws_lmax + (size_t)split * a.ws_rows;
        osum = a.ws_lsum + (size_t)split * a.ws_rows;
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = qblock * ROWS + qb * 32 + li;
        const float l_run = qb == 0 ? l_run0 : l_run1;
        const float m_run = qb == 0 ? m_run0 : m_run1;
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        if (qrow < a.m) {
            float *orow = out + (size_t)qrow * ldo + dvw0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col0 = NT * crow(r, hi);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
                    if (dvw0 + col0 + tt < a.dv) orow[col0 + tt] = oacc[tt][qb][r];
            }
            if (wave == 0 && hi == 0 && chunk == 0) {
                omax[qrow] = m_run * scale;
                osum[qrow] = l_tot;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host-side launch logic
// ---------------------------------------------------------------------------
// per-wave dv slice (a dv chunk = 4 slices)
// per-wave dv slice (a dv chunk = 4 slices), matched to dv so that no P.V MFMA runs on padding columns and a
// second chunk -- which recomputes the scores -- only starts where one workgroup's accumulators end: two query
// blocks (dk <= 512) hold 128 columns per wave, one block (dk > 512) 256
static inline int dksplit_slice(int dk, int dv) {
    if (dv <= 128) return 32;
    if (dv <= 256) return 64;
    if (dv <= 384) return 96;
    if (dv <= 512 || dk <= 512) return 128;
    return dv <= 768 ? 192 : 256;
}
int dksplit_chunks(int dk, int dv) { return (dv + 4 * dksplit_slice(dk, dv) - 1) / (4 * dksplit_slice(dk, dv)); }
int dksplit_rows(int dk) { return dk > 512 ? 32 : 64; }      // query rows per workgroup: one block beyond dk = 512

template <int DKS, int DVS, int QB>
static hipError_t launch_one(const PartialArgs &a, hipStream_t s) {
    const int nqb = (a.m + 32 * QB - 1) / (32 * QB);
    const int ntiles = (a.n_local + kKvTile - 1) / kKvTile;
    const int tiles_per_split = (ntiles + a.kv_splits - 1) / a.kv_splits;
    const int kv_per_split = tiles_per_split > 0 ? tiles_per_split * kKvTile : kKvTile;
    const int chunks = (a.dv + 4 * DVS - 1) / (4 * DVS);
    const size_t lds = (size_t)2 * 4 * QB * 64 * 20 * sizeof(float);
    static std::atomic<bool> attr_done[64];   // (zero-initialised; set from any enqueue thread)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_dksplit_pipe_kernel<DKS, DVS, QB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[dev] = true;
    }
    const float scale = 1.0f / sqrtf((float)a.dk);   // attention-mpi.c:208
    hipLaunchKernelGGL((fused_dksplit_pipe_kernel<DKS, DVS, QB>), dim3(nqb * chunks * a.kv_splits), dim3(256), lds, s,
                       a, kv_per_split, nqb, chunks, scale);
    note_launch("fused_dksplit_pipe_kernel", 3, DKS, DVS, QB, 0, 0,
                nqb * chunks * a.kv_splits, a.kv_splits, 0, a.m, a.n_local);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.kv_splits > 1 && !a.defer_merge) e = launch_split_merge(a, s);
    return e;
}

// the slice a launch uses: the matched one when the V image's rows are a whole number of lane runs (dense_ld()
// images always are), else the 128-wide one, whose runs of 4 fit every leading dimension the API accepts
static inline int launch_slice_width(const PartialArgs &a) {
    const int w = dksplit_slice(a.dk, a.dv);
    return (a.ldv % (w / 32) == 0) ? w : 128;
}

template <int DKS, int QB>
static hipError_t launch_slice(const PartialArgs &a, hipStream_t s) {
    switch (launch_slice_width(a)) {
        case 32: return launch_one<DKS, 32, QB>(a, s);
        case 64: return launch_one<DKS, 64, QB>(a, s);
        case 96: return launch_one<DKS, 96, QB>(a, s);
        case 128: return launch_one<DKS, 128, QB>(a, s);
        default:
            if constexpr (QB == 1) {
                if (launch_slice_width(a) == 192) return launch_one<DKS, 192, 1>(a, s);
                return launch_one<DKS, 256, 1>(a, s);
            } else {
                return hipErrorInvalidValue;      // two query blocks never get more than 128 columns per wave
            }
    }
}

hipError_t launch_dksplit(const PartialArgs &a, hipStream_t s) {
    if (a.dk > 768) return launch_slice<256, 1>(a, s);       // 768 < dk <= 1024: 256-wide dk slices, ONE query block per workgroup
    if (a.dk > 512) return launch_slice<192, 1>(a, s);       // 512 < dk <= 768: 192-wide slices, one block
    if (a.dk > 384) return launch_slice<128, 2>(a, s);
    if (a.dk > kMaxMfmaDk) return launch_slice<96, 2>(a, s); // 256 < dk <= 384: 96-wide slices, no score MFMAs on padding
    // non-dense 128 < dk <= 256 with dv > 128
    switch (launch_slice_width(a)) {
        case 64: return launch_one<64, 64, 2>(a, s);
        case 96: return launch_one<64, 96, 2>(a, s);
        default: return launch_one<64, 128, 2>(a, s);
    }
}

void dma_audit_read_dksplit(unsigned long long out[2]) {
    out[0] = out[1] = 0;
#ifdef SDPA_DMA_ASSERT
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dks_audit), 2 * sizeof(unsigned long long));
#endif
}

// (sdpa_internal.h: preload_kernels_*) touching one kernel makes the runtime load this translation unit's code object for the
// current device NOW -- not in front of the first launch that needs it, possibly behind a resident persistent launch
hipError_t preload_kernels_dksplit() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&fused_dksplit_pipe_kernel<128, 128, 2>));
}

}  // namespace sdpa
