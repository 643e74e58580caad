// sdpa_fwd_bf16.hip -- bf16-input MFMA variant of the fused online-softmax kernel
// (BASELINE.json config 5: d_k = d_v up to 512, fp32 accumulate and softmax).
//
// Same hot path as sdpa_fwd_f32.hip (online_softmax_attention, attention-mpi.c:168-189,
// for all rows of a Q batch against one K/V shard; same (contrib, lmax, lsum) outputs), with
// the operands rounded to bf16 instead of fp32 (the reference's mixed-precision step,
// attention-mpi.c:31-64, taken one notch further) and the two contractions on
// v_mfma_f32_32x32x16_bf16 (dense peak ~2.5 PFLOP/s).  Tolerance of this path:
// max|delta| <= 1e-2 * max(1, max|V|).
//
// Layout decisions specific to this variant:
//   * V is handed over TRANSPOSED: Vt[dv][n] bf16 (the convert kernel writes it that way).  The
//     P.V MFMA needs, per lane, kv-consecutive elements of one V column; with Vt that is an
//     8-byte LDS read, no transpose instruction and no shuffle.
//   * k-slot mapping of the second MFMA is chosen to match the C/D layout of the first: step s
//     of half-wave hi uses key rows 16s + {4hi..4hi+3, 8+4hi..8+4hi+3}, which are exactly the
//     rows accumulator registers 8s..8s+7 of that lane hold -- P goes from the softmax
//     registers into the B operand with a bf16 pack only.
//   * so that those eight keys are 16 CONTIGUOUS bytes of a Vt row, the image stores key j of a
//     row at position kvpos(j) = j with bits 2 and 3 swapped (each 16-key group is laid out
//     0-3, 8-11, 4-7, 12-15).  The convert kernel writes it that way; bf16_kvpos() is the map.
//   * d = 512 does not fit one wave's registers as a 32x512 fp32 O tile next to a 32x512 Q
//     fragment twice over, so dv is processed in chunks of <= 256 columns by separate
//     workgroups (blockIdx carries the chunk); the score tile is recomputed per chunk.
//   * workgroup = 4 waves x 32 query rows, 32-row K/V tiles, register-staged double buffer,
//     one barrier per tile (the structure of fused_partial_kernel).
#include "sdpa_internal.h"
#include "sdpa_debug.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

SDPA_AUDIT_COUNTER(g_bf16_audit)

// -DSDPA_BF16_MASK_EVERY_STEP=1: the ragged-tile mask in every step again (rounds 1-3), for the A/B of its cost
#ifndef SDPA_BF16_MASK_EVERY_STEP
#define SDPA_BF16_MASK_EVERY_STEP 0
#endif

namespace sdpa {

#ifdef SDPA_DMA_ASSERT
// a DMA source of 16 bytes must lie inside the K image or inside the Vt image of the launch
__device__ inline void bf16_audit_src(const Bf16Args &a, const char *src) {
    // (tiled images hold whole 32-key tiles of K rows; both Vt layouts hold padded dv x padded keys elements)
    const size_t npad = ((size_t)a.n_local + 31) / 32 * 32;
    const char *k0 = reinterpret_cast<const char *>(a.K), *k1 = k0 + (a.tiled ? npad : (size_t)a.n_local) * a.ldk * 2;
    const int ch = a.dv <= 64 ? 64 : a.dv <= 128 ? 128 : a.dv <= 256 ? 256 : 512;
    const char *v0 = reinterpret_cast<const char *>(a.Vt), *v1 = v0 + (size_t)((a.dv + ch - 1) / ch * ch) * (a.tiled ? npad : (size_t)a.ldvt) * 2;
    const bool in_k = src >= k0 && src + 16 <= k1, in_v = src >= v0 && src + 16 <= v1;
    if (!(in_k || in_v) || (reinterpret_cast<unsigned long long>(src) & 15ull) != 0) atomicAdd(&g_bf16_audit[0], 1ull);
}
#define SDPA_BF16_AUDIT(a, src) bf16_audit_src(a, src)
#else
#define SDPA_BF16_AUDIT(a, src) ((void)0)
#endif


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ constexpr int crow16(int r, int hi) {
    return (r & 3) + 8 * (r >> 2) + 4 * hi;
}

__device__ __forceinline__ unsigned f32_to_bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);          // round to nearest even (finite inputs)
    return u >> 16;
}

// two fp32 -> one dword of two bf16 (RNE) in ONE instruction; gfx950 has v_cvt_pk_bf16_f32 but
// hipcc exposes no builtin for it
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ int xcd_remap_b(int bid, int total) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = total >> 3, r = total & 7;
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return first + slot;
}

// ---------------------------------------------------------------------------
// Software-pipelined variant (the fp32 kernel's recipe, sdpa_fwd_f32.hip, for bf16 operands):
//   * K tiles go global -> LDS by LDS-DMA from inline asm (no staging registers, no ds_write),
//     16-byte chunks XOR-swizzled (chunk ^ (row & 15)) on the source side and undone by the
//     fragment reads; K is staged two tiles ahead in two buffers.  Vt tiles (64-byte rows) keep
//     the register path into padded 72-byte rows: a lane-linear DMA image of them would make every
//     ds_read_b64 of the P.V operand 2-way conflicted.
//   * two score tiles are live: S^T(t+1) = K(t+1).Q^T runs on the matrix pipe while S(t) becomes
//     P(t) (fma, exp2, bf16 pack) on the VALU -- which, unlike for f32-input MFMA, really does
//     run beside the bf16 MFMA -- and the row max of S(t+1) is reduced under the P.V MFMAs.
//   * the reference exponent m_ref only moves when a tile's max rises by more than 2^24; the
//     true row max is folded back at the epilogue (same scheme as the fp32 pipelined kernel).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float bpin_exp2(float x) {
    float y;
    asm volatile("v_exp_f32 %0, %1\n\ts_nop 0" : "=v"(y) : "v"(x));
    return y;
}
__device__ __forceinline__ unsigned bpin_pack(float lo, float hi) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bpin_max3(float a, float b, float c) {
    float y;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(y) : "v"(a), "v"(b), "v"(c));
    return y;
}

// fragment prefetch depths of the one-wave-per-SIMD shapes (K ring, V ring)
#ifndef SDPA_BF16_KD1
#define SDPA_BF16_KD1 3
#endif
#ifndef SDPA_BF16_VD1
#define SDPA_BF16_VD1 2
#endif

template <int DK, int DVC, int ABL = 0>
__global__ __launch_bounds__(256, (DK + 2 * DVC > 512) ? 1 : 2) void fused_bf16_pipe_kernel(
    Bf16Args a, int kv_per_split, int n_qblocks, int n_chunks, float scale) {
    SDPA_AUDIT_LAUNCH(g_bf16_audit);
    constexpr int NKS = DK / 16;               // QK^T k-steps = MFMAs per score tile
    constexpr int NT = DVC / 32;
    constexpr int KCH = DK / 8;                // 16-byte chunks per K row
    constexpr int KTILE = kKvTile * DK;        // bf16 elements, unpadded (swizzled)
    constexpr int VLD = 36;
    constexpr int VTILE = DVC * VLD;
    constexpr int KPW = (kKvTile * KCH / 64) / 4;   // 1-KiB DMA pieces per wave per K tile
    constexpr int RPP = 64 / KCH > 0 ? 64 / KCH : 1; // K rows per DMA piece (1 at DK = 512)
    constexpr int VPT = DVC / 64;
    constexpr int SWZ = KCH >= 16 ? 15 : KCH - 1;
    static_assert(KCH >= 8 && KPW >= 1, "DK must be 64..512");

    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short *const Ks = smem16;                    // [2][KTILE]
    unsigned short *const Vs = smem16 + 2 * KTILE;        // [2][VTILE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int hi = lane >> 5;

    int work = xcd_remap_b(blockIdx.x, gridDim.x);
    const int qblock = work % n_qblocks;
    work /= n_qblocks;
    const int chunk = work % n_chunks;
    const int split = work / n_chunks;
    const int qrow = qblock * kQRowsPerBlock + wave * 32 + li;
    const int dv0 = chunk * DVC;
    // second pass behind the wide kernel: only the blocks it flagged
    if (a.redo != nullptr && a.redo[split * n_qblocks + qblock] != a.redo_gen) return;

    const int kv_begin = split * kv_per_split;
    const int kv_end = min(a.n_local, kv_begin + kv_per_split);
    const int T = kv_end > kv_begin ? (kv_end - kv_begin + kKvTile - 1) / kKvTile : 0;
    // the Q image is pre-multiplied by log2(e)/sqrtf(dk) by its converter (sdpa_dev_cvt_d2bf_q), so
    // the MFMA chains deliver exp2-domain scores: the multiplier of the softmax argument is 1
    (void)scale;
    constexpr float c = 1.0f;

    // in the 512-register mode every MFMA result lands in the accumulator file, so one score
    // tile (16) sits there beside O: that many Q registers stay on the VGPR side
    constexpr int kQfInVgpr = 4;
    u32x4 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (qrow < a.m)
            qf[ks] = *reinterpret_cast<const u32x4 *>(a.Q + (size_t)qrow * DK + 16 * ks + 8 * hi);
        else
            qf[ks] = u32x4{0u, 0u, 0u, 0u};
    }
    // In the one-wave-per-SIMD shapes (d = 512) re-define the Q fragments as ACCUMULATOR-file
    // values: an empty asm with an "a" constraint makes them AGPR-class virtual registers, and an
    // MFMA may read its A/B operands straight from AGPRs -- so the 128 wave-persistent Q registers
    // sit beside the O accumulators instead of filling the VGPR half that prefetch needs.
    if constexpr (DK + 2 * DVC > 512) {
#pragma unroll
        for (int ks = 0; ks < NKS - kQfInVgpr; ++ks) asm volatile("" : "+a"(qf[ks]));
    }

    f32x16 oacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    constexpr float kDeferLog2 = 24.0f;
    float m_ref = 0.f, max_rel = 0.f, l_run = 0.f;       // exp2 domain, see the fp32 kernel

    // ---- K staging by LDS-DMA
    // (as the redo pass of a dv > 256 shape -- DVC = 256 -- this kernel reads the TILED images, sdpa_internal.h: the K rows carry
    //  their chunk swizzle already, the Vt tiles are contiguous blocks)
    const bool tiled = DVC == 256 && a.tiled != 0;
    unsigned koff[KPW];
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int row = (wave * KPW + j) * RPP + lane / KCH;
        const int cpos = lane % KCH;
        koff[j] = (unsigned)(row * DK * 2 + ((tiled ? cpos : (cpos ^ (row & SWZ))) << 4));
    }
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)smem16);
    auto dma_piece = [&](const char *gbase, unsigned lane_off, unsigned lds_byte) __attribute__((always_inline)) {
        SDPA_BF16_AUDIT(a, gbase + lane_off);
        if constexpr (ABL & 1) return;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %2\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane_off), "s"(lds_byte), "s"(gbase)
                     : "memory");
    };
    auto dma_k = [&](int tile, int buf) __attribute__((always_inline)) {
        const int base = kv_begin + tile * kKvTile;
        const int last = kv_end - 1 - base;
        const char *kb = reinterpret_cast<const char *>(a.K + (size_t)base * DK);
        if (last >= kKvTile - 1) {
#pragma unroll
            for (int j = 0; j < KPW; ++j)
                dma_piece(kb, koff[j], lds_base + (unsigned)(buf * KTILE * 2 + (wave * KPW + j) * 1024));
        } else {
#pragma unroll
            for (int j = 0; j < KPW; ++j) {
                const unsigned row = min((int)(koff[j] / (DK * 2)), last);
                dma_piece(kb, row * (DK * 2) + (koff[j] % (DK * 2)),
                          lds_base + (unsigned)(buf * KTILE * 2 + (wave * KPW + j) * 1024));
            }
        }
    };
    // ---- Vt staging through registers
    u32x4 vreg[VPT];
    unsigned voff[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = tid + 256 * i;
        const int vrow = dv0 + idx / 4, vch = idx % 4;
        voff[i] = tiled ? (unsigned)(((size_t)(vrow >> 9) * (512 * kKvTile) + (size_t)(vrow & 511) * kKvTile + ((vch ^ ((vrow >> 2) & 3)) << 3)) * 2u)
                        : (unsigned)(((size_t)vrow * a.ldvt + 8 * vch) * 2u);
    }
    const size_t vtile_stride = (size_t)((a.dv + 511) / 512 * 512) * kKvTile;      // tiled: elements between two key tiles
    auto v_gload = [&](int tile) __attribute__((always_inline)) {
        if constexpr (ABL & 1) return;
        const char *vb = tiled ? reinterpret_cast<const char *>(a.Vt + (size_t)(kv_begin / kKvTile + tile) * vtile_stride)
                               : reinterpret_cast<const char *>(a.Vt + kv_begin + tile * kKvTile);
#pragma unroll
        for (int i = 0; i < VPT; ++i) vreg[i] = *reinterpret_cast<const u32x4 *>(vb + voff[i]);
    };
    auto v_lstore = [&](int buf) __attribute__((always_inline)) {
        if constexpr (ABL & 1) return;
        unsigned short *vd = Vs + buf * VTILE;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = tid + 256 * i;
            unsigned short *dst = vd + (idx / 4) * VLD + 8 * (idx % 4);
            *reinterpret_cast<u32x2 *>(dst) = u32x2{vreg[i].x, vreg[i].y};
            *reinterpret_cast<u32x2 *>(dst + 4) = u32x2{vreg[i].z, vreg[i].w};
        }
    };
    auto stage_fence = [&]() __attribute__((always_inline)) {
        if constexpr (ABL & 8) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // K fragment byte addresses inside a K buffer (chunk 2ks+hi of row li, un-swizzled)
    constexpr int NKA = NKS < 8 ? NKS : 8;
    unsigned kaddr[NKA];
#pragma unroll
    for (int u = 0; u < NKA; ++u) kaddr[u] = (unsigned)(li * DK * 2 + (((2 * u + hi) ^ (li & SWZ)) << 4));
    auto kfrag = [&](int buf, int ks) __attribute__((always_inline)) -> u32x4 {
        if constexpr (ABL & 2) return qf[(ks + 1) % NKS];
        return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(Ks + buf * KTILE) +
                                                kaddr[ks % NKA] + (ks / NKA) * 256);
    };
    auto mask_ragged = [&](f32x16 &sx, int tile) __attribute__((always_inline)) {
        const int valid = kv_end - (kv_begin + tile * kKvTile);
        if (valid < kKvTile) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (crow16(r, hi) >= valid) sx[r] = -INFINITY;
        }
    };
    // scores in sx are raw dots; rel(s) = s*c - m_ref.  Folds a tile's max into the state.
    auto absorb_rel = [&](float tmax_raw) __attribute__((always_inline)) {
        float tmax = fmaf(tmax_raw, c, -m_ref);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        if (__any(tmax > kDeferLog2)) {
            const float jump = fmaxf(tmax, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-jump);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[tt][r] *= alpha;
            l_run *= alpha;
            m_ref += jump;
            max_rel -= jump;
            tmax -= jump;
        }
        max_rel = fmaxf(max_rel, tmax);
    };

    // One wave per SIMD: keep the O accumulators AGPR-class across the step boundary.  Left to
    // itself hipcc carries them as VGPR values (the rare rescale multiplies them) and shuttles all
    // of them through v_accvgpr_read/write around the P.V MFMAs of EVERY step.
    auto pin_o = [&]() __attribute__((always_inline)) {
        if constexpr (DK + 2 * DVC > 512) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) asm volatile("" : "+a"(oacc[tt]));
        }
    };

    // ... and the score tiles VGPR-class (the softmax reads every element): with Q and O the
    // accumulator file is exactly full, a score tile parked there pushes Q fragments to scratch.
    auto pin_s = [&](f32x16 &sx) __attribute__((always_inline)) {
        if constexpr (DK + 2 * DVC > 512) asm volatile("" : "+v"(sx));
    };

    constexpr int PPK = NKS >= 16 ? 1 : 16 / NKS;        // P values finished per QK^T MFMA slot
    constexpr int KPP = NKS >= 16 ? NKS / 16 : 1;        // QK^T MFMAs per P value

    auto step = [&](auto has_next, auto masked, f32x16 &su, f32x16 &sm, int t) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(has_next)::value;
        constexpr bool MASKED = decltype(masked)::value;   // this step scores the shard's last (possibly ragged) tile
        const int vbuf = t & 1, kbuf = (t + 1) & 1;
        if (t + 2 < T) dma_k(t + 2, t & 1);
        if (t + 1 < T) v_gload(t + 1);
        u32x4 pb[2];
        unsigned pw[8];

        auto p_slice = [&](int r0, int cnt) __attribute__((always_inline)) {
#pragma unroll
            for (int r = r0; r < r0 + cnt; ++r) {
                su[r] = bpin_exp2(fmaf(su[r], c, -m_ref));
                l_run += su[r];
                if (r & 1) pw[r >> 1] = bpin_pack(su[r - 1], su[r]);
            }
        };

        if constexpr (HAS_NEXT) {
            // [A] S^T(t+1) on the matrix pipe  ||  P(t) on the VALU
            // K fragments are read KD steps ahead of their MFMA (a bf16 MFMA retires in 32 cycles, an
            // LDS read takes ~4x that; deeper rings measured slower: they spill inside the loop)
            constexpr int KD = NKS < 6 ? NKS : (DK + 2 * DVC > 512 ? SDPA_BF16_KD1 : 2);
            u32x4 kq[KD];
#pragma unroll
            for (int i = 0; i < KD; ++i) kq[i] = kfrag(kbuf, i);
#pragma unroll
            for (int r = 0; r < 16; ++r) sm[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const u32x4 kf = kq[ks % KD];
                __builtin_amdgcn_sched_barrier(0);
                sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf),
                                                             __builtin_bit_cast(bf16x8, qf[ks]), sm, 0, 0, 0);
                if (ks + KD < NKS) kq[ks % KD] = kfrag(kbuf, ks + KD);
                if (ks % KPP == 0) p_slice((ks / KPP) * PPK, PPK);
            }
            __builtin_amdgcn_sched_barrier(0);
            pin_s(sm);
        } else {
            p_slice(0, 16);
        }
        pb[0] = u32x4{pw[0], pw[1], pw[2], pw[3]};
        pb[1] = u32x4{pw[4], pw[5], pw[6], pw[7]};

        // [B] O^T += Vt(t).P(t)^T on the matrix pipe  ||  row max of S^T(t+1), Vt(t+1) -> LDS
        if constexpr (HAS_NEXT && (MASKED || SDPA_BF16_MASK_EVERY_STEP)) mask_ragged(sm, t + 1);
        const unsigned short *vt = Vs + vbuf * VTILE + li * VLD + 8 * hi;
        float tmax = -INFINITY;
        constexpr int SLOTS = 2 * NT;                     // P.V MFMAs of this step
        constexpr int VD = (DK + 2 * DVC > 512) ? SDPA_BF16_VD1 : 2;   // V fragment prefetch depth
        auto vfrag = [&](int slot) __attribute__((always_inline)) -> u32x4 {
            const int h = slot / NT, tt = slot % NT;
            if constexpr (ABL & 2) return qf[(tt + h) % NKS];
            const unsigned short *vp = vt + (32 * tt) * VLD + 16 * h;
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(vp);        // keys 16h+4hi .. +3
            const u32x2 up = *reinterpret_cast<const u32x2 *>(vp + 4);    // keys 16h+8+4hi .. +3
            return u32x4{lo.x, lo.y, up.x, up.y};
        };
        u32x4 vq[VD];
#pragma unroll
        for (int i = 0; i < VD; ++i) vq[i] = vfrag(i);
#pragma unroll
        for (int slot = 0; slot < SLOTS; ++slot) {
            const int h = slot / NT, tt = slot % NT;
            const u32x4 vf = vq[slot % VD];
            __builtin_amdgcn_sched_barrier(0);
            oacc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf),
                                                               __builtin_bit_cast(bf16x8, pb[h]),
                                                               oacc[tt], 0, 0, 0);
            if (slot + VD < SLOTS) vq[slot % VD] = vfrag(slot + VD);
            if constexpr (HAS_NEXT) {
                // 16 maxes spread over the MFMA slots, starting one slot late
                if (slot >= 1) {
                    constexpr int per = (16 + SLOTS - 2) / (SLOTS - 1);
                    const int r0 = (slot - 1) * per;
#pragma unroll
                    for (int r = r0; r < r0 + per && r < 16; r += 2)
                        tmax = (r + 1 < 16 && r + 1 < r0 + per) ? bpin_max3(tmax, sm[r], sm[r + 1])
                                                                : bpin_max3(tmax, sm[r], sm[r]);
                }
            }
            if (slot == NT - 1 && t + 1 < T) v_lstore((t + 1) & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        pin_o();
        if constexpr (HAS_NEXT) absorb_rel(tmax);
        pin_o();
        stage_fence();
    };

    f32x16 sA, sB;
    if (T > 0) {
        dma_k(0, 0);
        v_gload(0);
        if (T > 1) dma_k(1, 1);
        v_lstore(0);
        stage_fence();
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const u32x4 kf = kfrag(0, ks);
            sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf),
                                                         __builtin_bit_cast(bf16x8, qf[ks]), sA, 0, 0, 0);
        }
        mask_ragged(sA, 0);
        float tmax = sA[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sA[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        m_ref = tmax * c;                               // finite: every tile has a valid key row
        __syncthreads();                                // K(0) fully consumed before K(2) lands on it

        int t = 0;
        using yes = std::true_type;
        using no = std::false_type;
        // Only the shard's LAST tile can be ragged, so only the step that scores it masks: the steady-state loop
        // stops two tiles short of the end and the tail below runs the masking instantiation.  (Round 4: with the
        // mask in every step hipcc turned its wave-uniform branch into ~45 selects per step, a third of the loop's
        // VALU work, next to MFMAs that do not overlap with VALU issue -- profiles/r04/bf16_ragged_mask_hoist.log.)
        for (; t + 3 < T; t += 2) {
            step(yes(), no(), sA, sB, t);
            step(yes(), no(), sB, sA, t + 1);
        }
        if (T - t == 3) {
            step(yes(), no(), sA, sB, t);
            step(yes(), yes(), sB, sA, t + 1);
            step(no(), no(), sA, sB, t + 2);
        } else if (T - t == 2) {
            step(yes(), yes(), sA, sB, t);
            step(no(), no(), sB, sA, t + 1);
        } else {
            step(no(), no(), sA, sB, t);
        }
    }

    // ---- epilogue: fold the true row max back in, write this chunk's columns
    const float fold = __builtin_amdgcn_exp2f(-max_rel);
    const float l_tot = (l_run + __shfl_xor(l_run, 32)) * fold;
    float *out = a.contrib;
    float *omax = a.lmax, *osum = a.lsum;
    int ldo = a.ldo;
    if (a.kv_splits > 1) {
        ldo = a.ws_ld;
        out = a.ws_contrib + (size_t)split * a.ws_rows * ldo;
        omax = a.ws_lmax + (size_t)split * a.ws_rows;
        osum = a.ws_lsum + (size_t)split * a.ws_rows;
    }
    if (qrow < a.m) {
        float *orow = out + (size_t)qrow * ldo + dv0;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = 32 * tt + crow16(r, hi);
                if (dv0 + col < a.dv) orow[col] = oacc[tt][r] * fold;
            }
        if (hi == 0 && chunk == 0) {
            omax[qrow] = T > 0 ? (m_ref + max_rel) * 0.69314718055994530942f : -INFINITY;
            osum[qrow] = l_tot;
        }
    }
}

// ---------------------------------------------------------------------------
// dv > 256 (BASELINE config 5): the tandem kernel, sdpa_fwd_bf16_tandem.inc (included below, twice).  One wave per
// SIMD; the pair's 64 x 256 fp32 O^T tiles fill the accumulator file (all 256 AGPRs), so the value columns of a
// 512-wide chunk are ONE pass and S^T is computed once.  What its text rests on:
//   * the 512-register mode of hipcc puts every MFMA-builtin result in AGPRs; with O filling them the score tile
//     must be a VGPR value, so the QK^T chain is issued as inline-asm MFMAs with VGPR C/D operands (first link
//     with the inline constant 0 as C).  The compiler does not see an MFMA in an asm statement and inserts no
//     MFMA->VALU wait states for it: the first VALU read of the score tile is placed >= 8 MFMAs behind the last
//     link, or behind an explicit s_nop pair (prologue, ragged mask, tail steps).
//   * O is touched by nothing but MFMAs between the zero-fill and the epilogue.  Any VALU access to those 256
//     registers in or around the loop -- a rescale on a never-taken branch, an early exit -- makes hipcc copy and
//     spill accumulator tiles on the hot path.  So the kernel has NO rescale and no row max: P = 2^score, a row whose
//     sum leaves [2^-80, 2^80] stamps its (q block, split) flag with the launch's generation number (nothing to
//     clear beforehand), and the launcher runs the general kernel (256-column chunks, in-loop rescale) behind this
//     one over the flagged blocks only.
//   * K and Vt (two LDS buffers each) arrive by LDS-DMA from the TILED images (sdpa_internal.h), issued piece by
//     piece between MFMAs.  Vt rows are 64 bytes per tile (32 keys); 16-byte chunk c of dv row r sits at position
//     c ^ ((r >> 2) & 3), which makes the P.V operand read -- one ds_read_b128 per lane per MFMA thanks to the
//     kvpos() key order -- conflict-free across each 16-lane group.
//   * the ceiling: the MFMA skeleton alone (no LDS, no DMA, no VALU) runs at 1.65-1.87 PFLOP/s with toggling operands
//     (tools/probes/mfma_probe.hip: 2.2 with constant ones) -- the part clocks to its power budget.
// ---------------------------------------------------------------------------
// hipcc does not see an MFMA inside an asm statement, so its hazard recogniser inserts none of the
// wait states an MFMA needs.  Two of them lead every statement: whatever VALU instruction the
// register allocator puts right in front of it to set up an operand (a v_accvgpr_write of a Q
// fragment, a v_mov of the accumulator) has then retired -- "VALU write VGPR/AGPR -> MFMA read"
// needs 2 wait states on gfx90a+ (found the hard way: a build that copied Q fragments into AGPRs
// in front of the tail steps' links read stale registers, nondeterministically).
#ifndef SDPA_MFMA_LEAD
#define SDPA_MFMA_LEAD "s_nop 1\n\t"
#endif
#ifndef SDPA_MFMA_ASM_TAIL      // tools/build_variant.sh: wait states BEHIND every asm MFMA (hazard hunting)
#define SDPA_MFMA_ASM_TAIL ""
#endif
__device__ __forceinline__ void mfma_bf16_vgpr_first(f32x16 &d, const u32x4 &x, const u32x4 &y) {
    asm volatile(SDPA_MFMA_LEAD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" SDPA_MFMA_ASM_TAIL : "=&v"(d) : "v"(x), "v"(y));
}
__device__ __forceinline__ void mfma_bf16_vgpr(f32x16 &d, const u32x4 &x, const u32x4 &y) {
    asm volatile(SDPA_MFMA_LEAD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" SDPA_MFMA_ASM_TAIL : "+v"(d) : "v"(x), "v"(y));
}
// the same with the B operand read from the ACCUMULATOR file (wave-persistent Q fragments parked there)
__device__ __forceinline__ void mfma_bf16_vgpr_first_qa(f32x16 &d, const u32x4 &x, const u32x4 &y) {
    asm volatile(SDPA_MFMA_LEAD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" SDPA_MFMA_ASM_TAIL : "=&v"(d) : "v"(x), "a"(y));
}
__device__ __forceinline__ void mfma_bf16_vgpr_qa(f32x16 &d, const u32x4 &x, const u32x4 &y) {
    asm volatile(SDPA_MFMA_LEAD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" SDPA_MFMA_ASM_TAIL : "+v"(d) : "v"(x), "a"(y));
}
// 16 wait states: covers an 8-pass MFMA's result latency before a VALU read (needs 11)
__device__ __forceinline__ void mfma_result_fence(f32x16 &d) {
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(d));
}
// max over the two half-waves without LDS: v_permlane32_swap exchanges lanes 32..63 of the first
// operand with lanes 0..31 of the second
__device__ __forceinline__ float halfwave_max(float x) {
    float lo = x, up = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(up));
    return fmaxf(lo, up);
}

// tandem kernel schedule knobs (tools/build_variant.sh A/Bs; the defaults are what profiles/r06/bf16_tiled_ab.log measured)
#ifndef SDPA_TANDEM_KD
#define SDPA_TANDEM_KD 3              // K fragment ring depth
#endif
#ifndef SDPA_TANDEM_VD
#define SDPA_TANDEM_VD 4              // Vt fragment ring depth
#endif
#ifndef SDPA_TANDEM_GROUP
#define SDPA_TANDEM_GROUP 2
#endif
#ifndef SDPA_TANDEM_XA
#define SDPA_TANDEM_XA 3              // chain links issued behind [B]'s first fragment reads
#endif
#ifndef SDPA_TANDEM_XB
#define SDPA_TANDEM_XB 4              // P.V MFMAs issued behind the step's barrier
#endif
// tools/build_variant.sh -DSDPA_TANDEM_ABL=bits: TIMING-ONLY ablations of the tandem kernel's steady-state loop (results
// are wrong), never in the shipped library: 1 = no softmax VALU (exp2 / row sum / pack), 2 = no LDS fragment reads
// (operands taken from the Q registers), 4 = no LDS-DMA issue in the loop (the prologue's tiles stay in LDS),
// 8 = no barriers, 16 = no P exchange.  -DSDPA_TANDEM_STAMP=1: s_memtime stamps per step, written over lsum (tools/gpu_bf16_ab.py)
#ifndef SDPA_TANDEM_ABL
#define SDPA_TANDEM_ABL 0
#endif
#ifndef SDPA_TANDEM_STAMP
#define SDPA_TANDEM_STAMP 0
#endif
// the steady-state chain's links carry no wait states: its operands are written by LDS reads (covered by s_waitcnt) and by
// the previous link only -- tests/test_kernel_isa.py checks that no VALU write of an operand sits right in front of one
__device__ __forceinline__ void mfma_bf16_vgpr_first_nl(f32x16 &d, const u32x4 &x, const u32x4 &y) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(x), "v"(y));
}
__device__ __forceinline__ void mfma_bf16_vgpr_nl(f32x16 &d, const u32x4 &x, const u32x4 &y) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(x), "v"(y));
}

#define SDPA_TD_STREAM 0
#include "sdpa_fwd_bf16_tandem.inc"
#undef SDPA_TD_STREAM
#define SDPA_TD_STREAM 1
#include "sdpa_fwd_bf16_tandem.inc"
#undef SDPA_TD_STREAM

// ---------------------------------------------------------------------------
// Duo variant (dk, dv <= 256): one wave per SIMD, TWO 32-row query blocks per wave.
// In the kernels above every K or Vt fragment read from LDS (1 KiB per wave instruction) feeds ONE
// MFMA, and four SIMDs issuing one v_mfma_f32_32x32x16_bf16 per 32 cycles then ask for exactly the
// 128 B/clk an LDS can deliver: the matrix pipe can never run ahead of the LDS.  Here a fragment is
// read once and multiplied against both query blocks of the wave, which halves LDS bytes per flop:
//   * workgroup = 4 waves x 64 query rows = 256 rows; per wave the two O^T tiles (2 x DV/32 x 16
//     registers) live in the ACCUMULATOR file, and so do the Q fragments (2 x DK/16 x 4) when both
//     fit its 256 registers (QA: 192 at d = 128), pinned there with "+a" constraints and read by the
//     MFMAs straight from it; the architectural VGPRs hold the softmax;
//   * SETS = 2 (d <= 128): two score-tile SETS are live (4 tiles, VGPR-form inline-asm MFMAs as in
//     the wide kernel):
//     [A] S^T(t+1) = K(t+1).Q^T for both blocks -- two independent chains interleaved, so a link
//         never waits for its predecessor -- and  [B] O^T += Vt(t).P(t)^T for both blocks;
//     the softmax of a tile (exp2, row sum, bf16 pack, row max: ~4 VALU per element pair) is cut
//     into 32 slices (block 0's 16 registers, then block 1's) issued ONE PER MFMA GAP across
//     [B] of step t-1 and [A] of step t, i.e. spread evenly under all 32 MFMAs of a step at d = 128;
//     SETS = 1 (dk = dv = 256: O fills the accumulator file, Q takes half the VGPRs): one set, all
//     32 slices under the 32 MFMAs of [B], the wide kernel's schedule;
//   * the Q image carries log2e/sqrt(dk) (its converter multiplies before the ONE bf16 rounding), so
//     the chains deliver exp2-domain scores and P = v_exp_f32(score): the reference exponent is
//     ZERO, no fma, no per-tile state.  The raw
//     row max is a running v_max3; after the last tile it is exchanged across the half-waves once,
//     becomes lmax, and a workgroup with a row whose scores left +-80 (log2 domain) is flagged and
//     redone by the general kernel right behind this one (in-loop rescale there);
//   * K (three buffers) and Vt (two) by LDS-DMA, issued piece by piece between MFMAs, one barrier
//     per step; image layouts and swizzles are the wide kernel's.
// ---------------------------------------------------------------------------
#ifndef SDPA_DUO_KD
#define SDPA_DUO_KD 3
#endif
#ifndef SDPA_DUO_VD
#define SDPA_DUO_VD 2
#endif
constexpr int kDuoRows = 256;          // query rows per workgroup of the duo kernel
// tools/build_variant.sh -DSDPA_DUO_ABL=bits: TIMING-ONLY ablations (results are wrong), never in
// the shipped library: 1 = no softmax VALU, 2 = no LDS fragment reads, 4 = no DMA, 8 = no barrier
#ifndef SDPA_DUO_ABL
#define SDPA_DUO_ABL 0
#endif

template <int DK, int DV>
struct DuoCfg {
    // Q fragments go to the accumulator file when they fit beside O with room to spare (a file filled
    // to the last register makes hipcc shuttle tiles through VGPRs inside the loop)
    static constexpr bool QA = DK / 2 + DV <= 192;
    static constexpr int SETS = (QA || DK <= 128) ? 2 : 1;      // live score-tile sets
    // LDS rings: K(t+NKB) is requested while K(t+1) is read, Vt(t+NVB-1) while Vt(t) is read.
    // Deeper rings (4..6 K buffers) measured the same +-1 % at d = 64..256 (profiles/r02/
    // bf16_duo_ring_depth_ab.log): the tiles come from L2 / Infinity Cache well inside one step.
#ifdef SDPA_DUO_NKB          // tools/build_variant.sh: ring depth A/B
    static constexpr int NKB = SDPA_DUO_NKB;
#else
    // (at dk = dv = 256 the VGPR file is exactly full and some ring depths tip hipcc into spilling a Q
    //  fragment inside the loop -- a scratch reload sits behind an s_waitcnt vmcnt(0) that also waits for
    //  the DMA in flight: +16..27 % -- tests/test_kernel_isa.py guards the property for every instantiation)
    static constexpr int NKB = 3;
#endif
    static constexpr int NVB = NKB - 1;
    static constexpr size_t lds_bytes = ((size_t)NKB * 32 * DK + (size_t)NVB * DV * 32) * 2;
};

#ifndef SDPA_DUO_PIN_IN_LOOP
#define SDPA_DUO_PIN_IN_LOOP 1
#endif


#define DUO_PIN_O() do { if constexpr (SDPA_DUO_PIN_IN_LOOP) pin_o(); } while (0)
template <int DK, int DV>
__global__ __launch_bounds__(256, 1) void fused_bf16_duo_kernel(
    Bf16Args a, int kv_per_split, int n_qblocks, int n_qblocks128, float scale) {
    SDPA_AUDIT_LAUNCH(g_bf16_audit);
    constexpr int NKS = DK / 16;               // QK^T k-steps: MFMAs per score tile and block
    constexpr int NT = DV / 32;                // 32-row blocks of O^T per query block
    constexpr int KCH = DK / 8;                // 16-byte chunks per K row
    constexpr int KTILE = kKvTile * DK;        // bf16 elements, unpadded (swizzled)
    constexpr int VTILE = DV * kKvTile;        // bf16 elements: 64-byte rows, swizzled
    constexpr int KPW = (kKvTile * KCH / 64) / 4;   // 1-KiB DMA pieces per wave per K tile
    constexpr int RPP = 64 / KCH > 0 ? 64 / KCH : 1; // K rows per DMA piece
    constexpr int VPW = (DV * 4 / 64) / 4;          // 1-KiB DMA pieces per wave per Vt tile (16 rows each)
    constexpr int SWZ = KCH >= 16 ? 15 : KCH - 1;
    constexpr int GA = 2 * NKS;                // MFMAs of phase [A]
    constexpr int GB = 4 * NT;                 // MFMAs of phase [B]
    constexpr int NSL = 32;                    // softmax slices per tile: 2 blocks x 16 accumulator registers
    constexpr int NKB = DuoCfg<DK, DV>::NKB, NVB = DuoCfg<DK, DV>::NVB;
    constexpr bool QA = DuoCfg<DK, DV>::QA;
    constexpr int SETS = DuoCfg<DK, DV>::SETS;
    // slices issued under [B]; the rest under the next step's [A] (none with one score set)
    constexpr int NB_SL = SETS == 1 ? NSL : NSL * GB / (GA + GB);
    static_assert(KCH >= 8 && KPW >= 1 && VPW >= 1, "DK, DV must be 64..256");

    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short *const Ks = smem16;                    // [NKB][KTILE]
    unsigned short *const Vs = smem16 + NKB * KTILE;      // [NVB][VTILE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int hi = lane >> 5;

    const int work = xcd_remap_b(blockIdx.x, gridDim.x);
    const int qblock = work % n_qblocks;
    const int split = work / n_qblocks;
    const int qrow0 = qblock * kDuoRows + wave * 64 + li;        // block b: qrow0 + 32 b

    const int kv_begin = split * kv_per_split;
    const int kv_end = min(a.n_local, kv_begin + kv_per_split);
    const int T = kv_end > kv_begin ? (kv_end - kv_begin + kKvTile - 1) / kKvTile : 0;
    (void)scale;      // log2(e)/sqrtf(dk) lives in the Q image (sdpa_dev_cvt_d2bf_q): scores arrive in the exp2 domain

    u32x4 qf[2][NKS];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int qrow = qrow0 + 32 * b;
            if (qrow < a.m)
                qf[b][ks] = *reinterpret_cast<const u32x4 *>(a.Q + (size_t)qrow * DK + 16 * ks + 8 * hi);
            else
                qf[b][ks] = u32x4{0u, 0u, 0u, 0u};
        }
    // wave-persistent operands live in the accumulator file: an MFMA reads A/B from AGPRs as well
    auto pin_q = [&]() __attribute__((always_inline)) {
        if constexpr (QA) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+a"(qf[b][ks]));
        }
    };
    // one link of a score chain: S^T += K fragment . Q fragment (ks-th k-step of block b)
    auto score_link = [&](f32x16 &sx, const u32x4 &kf, int b, int ks) __attribute__((always_inline)) {
        if constexpr (QA) {
            if (ks == 0) mfma_bf16_vgpr_first_qa(sx, kf, qf[b][0]);
            else mfma_bf16_vgpr_qa(sx, kf, qf[b][ks]);
        } else {
            if (ks == 0) mfma_bf16_vgpr_first(sx, kf, qf[b][0]);
            else mfma_bf16_vgpr(sx, kf, qf[b][ks]);
        }
    };
    pin_q();

    f32x16 oacc[2][NT];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[b][t][r] = 0.f;
    auto pin_o = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) asm volatile("" : "+a"(oacc[b][tt]));
    };
    pin_o();
    // |score * log2e / sqrt(dk)| <= 80 (|q.k/sqrt(dk)| <= 55), else the block is redone: P <= 2^80 leaves
    // 2^47 of fp32 headroom for sum_j P_j |V_j|
    float l_run[2] = {0.f, 0.f}, l_tot[2] = {0.f, 0.f};    // exp2 domain
    int fold_exp[2] = {0, 0};

    // ---- K and Vt staging by LDS-DMA (the wide kernel's scheme)
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)smem16);
    auto dma_piece = [&](const char *gbase, unsigned lane_off, unsigned lds_byte) __attribute__((always_inline)) {
        SDPA_BF16_AUDIT(a, gbase + lane_off);
        if constexpr (SDPA_DUO_ABL & 4) return;
        asm volatile("s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %2"
                     :
                     : "v"(lane_off), "s"(lds_byte), "s"(gbase)
                     : "memory" SDPA_M0_CLOBBER);
    };
    const unsigned klane = (unsigned)((lane / KCH) * DK * 2 + (((lane % KCH) ^ ((lane / KCH) & SWZ)) << 4));
    auto dma_k_piece = [&](int tile, int buf, int j) __attribute__((always_inline)) {
        const int base = kv_begin + tile * kKvTile;
        const int last = kv_end - 1 - base;
        const char *kb = reinterpret_cast<const char *>(a.K + (size_t)base * DK);
        const int row0 = (wave * KPW + j) * RPP;                  // wave-uniform, a multiple of RPP:
        const unsigned swz = (unsigned)((row0 & SWZ) << 4);       // (row0 + x) & SWZ == (row0 & SWZ) ^ x
        const unsigned dst = lds_base + (unsigned)(buf * KTILE * 2 + (wave * KPW + j) * 1024);
        // rows past the shard's end (only the last tile has any) re-read its last row: finite data,
        // their scores are masked.  (A scalar-only variant -- whole pieces moved back by a pointer fix, no
        // per-lane clamp -- cannot keep the valid rows of a straddling piece in place, and timed the same
        // on full tiles: profiles/r02/bf16_scalar_clamp_ab.log.)
        unsigned off;                                             // volatile: not hoisted into KPW live registers
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(off) : "s"(swz), "v"(klane));
        const unsigned row = (unsigned)min(row0 + (int)(lane / KCH), last);
        dma_piece(kb, row * (DK * 2) + (off % (DK * 2)), dst);
    };
    const unsigned vlane = (unsigned)((size_t)(lane >> 2) * a.ldvt * 2u) + (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    auto dma_v_piece = [&](int tile, int buf, int j) __attribute__((always_inline)) {
        const char *vb = reinterpret_cast<const char *>(a.Vt + kv_begin + tile * kKvTile);
        dma_piece(vb + (size_t)((wave * VPW + j) * 16) * a.ldvt * 2u, vlane,
                  lds_base + (unsigned)(NKB * KTILE * 2 + buf * VTILE * 2 + (wave * VPW + j) * 1024));
    };
    auto stage_fence = [&](auto keep) __attribute__((always_inline)) {
        constexpr int KEEP = decltype(keep)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
        if constexpr (!(SDPA_DUO_ABL & 8)) __syncthreads();
    };

    constexpr int NKA = NKS < 8 ? NKS : 8;
    unsigned kaddr[NKA];
#pragma unroll
    for (int u = 0; u < NKA; ++u) kaddr[u] = (unsigned)(li * DK * 2 + (((2 * u + hi) ^ (li & SWZ)) << 4));
    auto kfrag = [&](int ks) __attribute__((always_inline)) -> u32x4 {
        if constexpr (SDPA_DUO_ABL & 2) return u32x4{kaddr[ks % NKA], kaddr[0], kaddr[(ks + 1) % NKA], kaddr[ks % NKA]};
        return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(Ks) +
                                                kaddr[ks % NKA] + (ks / NKA) * 256);
    };
    unsigned vaddr[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) vaddr[h] = (unsigned)(li * 64 + (((2 * h + hi) ^ ((li >> 2) & 3)) << 4));
    // like kaddr[], vaddr[] carries the byte offset of the Vt buffer being read and is advanced in place
    auto vfrag = [&](int slot) __attribute__((always_inline)) -> u32x4 {
        const int h = slot / NT, tt = slot % NT;
        if constexpr (SDPA_DUO_ABL & 2) return u32x4{vaddr[h], kaddr[0], vaddr[0], (unsigned)tt};
        return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(Vs) + vaddr[h] + tt * 2048);
    };
    auto mask_ragged = [&](f32x16 (&sx)[2], int tile) __attribute__((always_inline)) {
        const int valid = kv_end - (kv_begin + tile * kKvTile);
        if (valid < kKvTile) {
            const int vh = valid - 4 * hi;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (crow16(r, 0) >= vh) sx[b][r] = -INFINITY;
        }
    };

    // ---- softmax slices.  Slice i = element r = i % 16 of block b = i / 16 of score set sx:
    //   e = exp2(s)  (issued now);  l += e, bf16 pack  (of the PREVIOUS slice: the transcendental's
    //   latency is never waited on).  Nothing else: the reference exponent is ZERO and no row max is
    //   tracked -- the row sum itself tells, after the last tile, whether the row stayed in range and
    //   which power of two to fold out (epilogue).  (Tried and measured slower on the same box,
    //   +6 % at d = 128 each: v_pk_add_f32 row sums, and v_dot2_f32_bf16 row sums of the packed
    //   weights -- profiles/r02/bf16_slice_variants_ab.log, bf16_dot2_rowsum_ab.log.)
    // State carried between slices of one tile: the two newest P values (element r-1: pending add and
    // pack partner; r-2: its pack partner).
    struct SliceState { float e0 = 0.f, e1 = 0.f; };
    auto slice = [&](int i, f32x16 (&sx)[2], SliceState &st, u32x4 (&pout)[2][2]) __attribute__((always_inline)) {
        const int b = i / 16, r = i % 16;
        if constexpr (SDPA_DUO_ABL & 1) {
            if (r == 15) {
                pout[b][0] = u32x4{__float_as_uint(sx[b][0]), __float_as_uint(sx[b][1]), __float_as_uint(sx[b][2]), __float_as_uint(sx[b][3])};
                pout[b][1] = u32x4{__float_as_uint(sx[b][4]), __float_as_uint(sx[b][5]), __float_as_uint(sx[b][6]), __float_as_uint(sx[b][7])};
            }
            return;
        }
        float e;
        asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(sx[b][r]));
        if (r > 0) {                                   // finish element r-1 of this block
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(l_run[b]) : "v"(st.e0));
            if (((r - 1) & 1) == 1) pout[b][((r - 1) >> 1) / 4][((r - 1) >> 1) % 4] = bpin_pack(st.e1, st.e0);
        }
        st.e1 = st.e0;
        st.e0 = e;
        if (r == 15) {                                 // close the block
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(l_run[b]) : "v"(st.e0));
            pout[b][1][3] = bpin_pack(st.e1, st.e0);
        }
    };
    // slices [lo, hi) spread over the gaps of a phase with G MFMAs: after MFMA j run those whose
    // index is below lo + ceil((j+1) (hi-lo) / G)
    auto slices_after = [&](int j, int G, int lo, int hi_, f32x16 (&sx)[2], SliceState &st,
                            u32x4 (&pout)[2][2]) __attribute__((always_inline)) {
        const int n = hi_ - lo;
        const int from = lo + (j * n + G - 1) / G, to = lo + ((j + 1) * n + G - 1) / G;
#pragma unroll
        for (int i = from; i < to; ++i) slice(i, sx, st, pout);
    };

    // rotating ring positions (wave-uniform): K buffer read / written in [A], Vt buffer read / written in [B]
    int kr = 1, kw = 0, vr = 0, vw = NVB - 1;
    // One step t.  scur: scores of tile t (its slices [NB_SL, 32) still to do), snxt: receives
    // S^T(t+1); pcur: P(t) (completed in [A]), pnxt: receives P(t+1) (slices [0, NB_SL) in [B]).
    auto step = [&](auto has_next, auto masked, f32x16 (&scur)[2], f32x16 (&snxt)[2], SliceState &stc, SliceState &stn,
                    u32x4 (&pcur)[2][2], u32x4 (&pnxt)[2][2], int t) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(has_next)::value;
        constexpr bool MASKED = decltype(masked)::value;   // this step scores the shard's last (possibly ragged) tile
        DUO_PIN_O();
#ifdef SDPA_DUO_PINQ
        pin_q();
#endif
        // [A]
        if constexpr (HAS_NEXT) {
            const int tk = min(t + NKB, T - 1);            // past the end: a harmless reload into a free buffer
            // K fragment ring: one less where Q already takes half the VGPRs (a fragment spilled to
            // scratch is reloaded behind an s_waitcnt vmcnt(0) that also waits for the DMA in flight)
            constexpr int KDW = SETS == 1 ? SDPA_DUO_KD - 1 : SDPA_DUO_KD;
            constexpr int KD = NKS < KDW ? NKS : KDW;
            u32x4 kq[KD];
#pragma unroll
            for (int i = 0; i < KD; ++i) kq[i] = kfrag(i);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const u32x4 kf = kq[ks % KD];
                __builtin_amdgcn_sched_barrier(0);
                score_link(snxt[0], kf, 0, ks);
                slices_after(2 * ks, GA, NB_SL, NSL, scur, stc, pcur);
                __builtin_amdgcn_sched_barrier(0);
                score_link(snxt[1], kf, 1, ks);
                if (ks + KD < NKS) kq[ks % KD] = kfrag(ks + KD);
                if (ks % 4 == 0) dma_k_piece(tk, kw, ks / 4);      // NKS / KPW == 4 for every DK
                slices_after(2 * ks + 1, GA, NB_SL, NSL, scur, stc, pcur);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_result_fence(snxt[0]);                            // the chains' last links have retired
            asm volatile("" : "+v"(snxt[1]));
            // everything older than Vt(t+1)'s request has landed: K(t+2) and Vt(t) with it
#ifdef SDPA_DUO_WAIT_ALL     // tools/build_variant.sh: debugging aid, drains every request at every step
            stage_fence(std::integral_constant<int, 0>());
#else
            stage_fence(std::integral_constant<int, (NVB - 1) * KPW + (NVB - 2) * VPW>());
#endif
        } else {
#pragma unroll
            for (int i = NB_SL; i < NSL; ++i) slice(i, scur, stc, pcur);
            stage_fence(std::integral_constant<int, 0>());
        }
        DUO_PIN_O();

        // [B]
        if constexpr (HAS_NEXT && (MASKED || SDPA_BF16_MASK_EVERY_STEP)) mask_ragged(snxt, t + 1);
        constexpr int SLOTS = 2 * NT;                      // Vt fragments of this step (2 MFMAs each)
        constexpr int VD = SDPA_DUO_VD;
        const int kr_next = kr == NKB - 1 ? 0 : kr + 1;
        const unsigned kstep = (unsigned)((kr_next - kr) * KTILE * 2);
        const int vr_next = vr == NVB - 1 ? 0 : vr + 1;
        const unsigned vstep = (unsigned)((vr_next - vr) * VTILE * 2);
        const int tv = min(t + NVB - 1, T - 1);            // past the end: a harmless reload
        u32x4 vq[VD];
#pragma unroll
        for (int i = 0; i < VD; ++i) vq[i] = vfrag(i);
#pragma unroll
        for (int slot = 0; slot < SLOTS; ++slot) {
            const int tt = slot % NT, h = slot / NT;
            const u32x4 vf = vq[slot % VD];
            __builtin_amdgcn_sched_barrier(0);
            oacc[0][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf),
                                                                  __builtin_bit_cast(bf16x8, pcur[0][h]),
                                                                  oacc[0][tt], 0, 0, 0);
            if constexpr (HAS_NEXT) slices_after(2 * slot, GB, 0, NB_SL, snxt, stn, pnxt);
            __builtin_amdgcn_sched_barrier(0);
            oacc[1][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf),
                                                                  __builtin_bit_cast(bf16x8, pcur[1][h]),
                                                                  oacc[1][tt], 0, 0, 0);
            if (slot + VD < SLOTS) vq[slot % VD] = vfrag(slot + VD);
            if constexpr (HAS_NEXT) {
                if (slot % 4 == 1) dma_v_piece(tv, vw, slot / 4);               // SLOTS / VPW == 4 for every DV
                slices_after(2 * slot + 1, GB, 0, NB_SL, snxt, stn, pnxt);
                // next step reads the next K buffer: the NKA fragment addresses advance, spread over the slots
#pragma unroll
                for (int u = slot * NKA / SLOTS; u < (slot + 1) * NKA / SLOTS; ++u) kaddr[u] += kstep;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        DUO_PIN_O();
        if constexpr (HAS_NEXT) {
            vaddr[0] += vstep;                              // all of this step's Vt reads are issued
            vaddr[1] += vstep;
            kr = kr_next;
            kw = kw == NKB - 1 ? 0 : kw + 1;
            vr = vr_next;
            vw = vw == NVB - 1 ? 0 : vw + 1;
        }
    };

    if (T > 0) {
        // fill the rings: K(0..NKB-1), Vt(0..NVB-2)
#pragma unroll
        for (int i = 0; i < NKB; ++i) {
#pragma unroll
            for (int j = 0; j < KPW; ++j) dma_k_piece(min(i, T - 1), i, j);
            if (i < NVB - 1) {
#pragma unroll
                for (int j = 0; j < VPW; ++j) dma_v_piece(min(i, T - 1), i, j);
            }
        }
        stage_fence(std::integral_constant<int, 0>());
        f32x16 sA[2], sB[2];
        u32x4 pA[2][2], pB[2][2];
        SliceState stA, stB;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const u32x4 kf = kfrag(ks);
            score_link(sA[0], kf, 0, ks);
            score_link(sA[1], kf, 1, ks);
        }
        mfma_result_fence(sA[0]);
        mfma_result_fence(sA[1]);
        mask_ragged(sA, 0);
#pragma unroll
        for (int i = 0; i < NB_SL; ++i) slice(i, sA, stA, pA);     // the part of P(0) a step's [B] would have done
        __syncthreads();                                // K(0) fully consumed before K(NKB) lands on it
#pragma unroll
        for (int u = 0; u < NKA; ++u) kaddr[u] += (unsigned)(KTILE * 2);   // step 0 reads K(1) in buffer 1

        int t = 0;
        using yes = std::true_type;
        using no = std::false_type;
        if constexpr (SETS == 2) {
            for (; t + 3 < T; t += 2) {             // (the last tile is scored by a masking step of the tail: see the pipe kernel)
                step(yes(), no(), sA, sB, stA, stB, pA, pB, t);
                step(yes(), no(), sB, sA, stB, stA, pB, pA, t + 1);
            }
            if (T - t == 3) {
                step(yes(), no(), sA, sB, stA, stB, pA, pB, t);
                step(yes(), yes(), sB, sA, stB, stA, pB, pA, t + 1);
                step(no(), no(), sA, sB, stA, stB, pA, pB, t + 2);
            } else if (T - t == 2) {
                step(yes(), yes(), sA, sB, stA, stB, pA, pB, t);
                step(no(), no(), sB, sA, stB, stA, pB, pA, t + 1);
            } else {
                step(no(), no(), sA, sB, stA, stB, pA, pB, t);
            }
        } else {            // one score set: it is dead once its slices ran under [B]
            for (; t + 3 < T; t += 2) {             // (the last tile is scored by a masking step of the tail: see the pipe kernel)
                step(yes(), no(), sA, sA, stA, stA, pA, pB, t);
                step(yes(), no(), sA, sA, stA, stA, pB, pA, t + 1);
            }
            if (T - t == 3) {
                step(yes(), no(), sA, sA, stA, stA, pA, pB, t);
                step(yes(), yes(), sA, sA, stA, stA, pB, pA, t + 1);
                step(no(), no(), sA, sA, stA, stA, pA, pB, t + 2);
            } else if (T - t == 2) {
                step(yes(), yes(), sA, sA, stA, stA, pA, pB, t);
                step(no(), no(), sA, sA, stA, stA, pB, pA, t + 1);
            } else {
                step(no(), no(), sA, sA, stA, stA, pA, pB, t);
            }
        }
        bool redo = false;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // reference exponent 0: P = 2^score.  max_j P_j <= l <= n max_j P_j, so a row sum inside
            // [2^-80, 2^80] means nothing overflowed (2^47 of fp32 headroom left for sum_j P_j |V_j|) and
            // the row did not flush to zero; a row outside -- or a NaN -- sends its workgroup to the redo pass
            l_tot[b] = l_run[b] + __shfl_xor(l_run[b], 32);
            const bool ok = l_tot[b] >= 0x1p-80f && l_tot[b] <= 0x1p80f;
            redo |= __any(!ok);
            // fold out the power of two just below the row sum: exact, and lsum lands in [1, 2)
            fold_exp[b] = ok ? __builtin_amdgcn_frexp_expf(l_tot[b]) - 1 : 0;
        }
        if (redo && lane == 0) {                        // flags are per 128-row block of the redo kernel
            const int q128 = 2 * qblock;
            a.redo[split * n_qblocks128 + q128] = a.redo_gen;
            if (q128 + 1 < n_qblocks128) a.redo[split * n_qblocks128 + q128 + 1] = a.redo_gen;
        }
    }

    // ---- epilogue: the triple relative to lmax = fold_exp * ln 2 (the reference exponent of this
    //      row: any value makes a valid triple for the merges; this one keeps lsum in [1, 2))
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
    for (int b = 0; b < 2; ++b) {
        const int qrow = qrow0 + 32 * b;
        if (qrow < a.m) {
            float *orow = out + (size_t)qrow * ldo;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = 32 * tt + crow16(r, hi);
                    if (col < a.dv) orow[col] = __builtin_amdgcn_ldexpf(oacc[b][tt][r], -fold_exp[b]);
                }
            if (hi == 0) {
                omax[qrow] = T > 0 ? (float)fold_exp[b] * 0.69314718055994530942f : -INFINITY;
                osum[qrow] = __builtin_amdgcn_ldexpf(l_tot[b], -fold_exp[b]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// converts: fp64 -> bf16 (RNE), row-major with zero-padded columns, and the transposed V image -- in the layout the
// shape's kernel reads (sdpa_internal.h: row images for dv <= 256, tiled images for dv > 256)
// ---------------------------------------------------------------------------
// The converters read their source ONCE: streaming (non-temporal) loads keep the fp64 arrays out of the Infinity Cache, where the
// operand images they write should stay for the fused kernel that follows -- measured on config 5 in bf16, same box: the tandem
// kernel 3.04 ms back to back, 3.21-3.23 behind plain-load converts (behind the Q convert alone: 134 MB of fp64 through the cache),
// and no slower than back to back behind streaming ones (profiles/r06/bf16_kernel_behind_converts.log).
#ifndef SDPA_CVT_NT
#define SDPA_CVT_NT 1
#endif
template <typename T> __device__ __forceinline__ T cvt_src_load(const T *p) {
#if SDPA_CVT_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// rows [0, rows) converted, rows [rows, rows_pad) zero; 16-byte chunk c of row r lands at chunk position c ^ (r & swz)
// (swz = 0: plain rows; the tiled K image: bf16_k_swz -- r counts from the image's first row, a multiple of 16 for `dst`)
__global__ void cvt_d2bf_kernel(const double *__restrict__ src, unsigned short *__restrict__ dst,
                                long rows, long rows_pad, int cols, int ld, double mult, int swz) {
    const long total = rows_pad * ld;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / ld;
        const int cidx = (int)(idx - r * ld);
        const unsigned short v = (r < rows && cidx < cols) ? (unsigned short)f32_to_bf16_rne(__double2float_rn(cvt_src_load(&src[r * cols + cidx]) * mult)) : (unsigned short)0;
        const int at = ((((cidx >> 3) ^ ((int)r & swz)) << 3) | (cidx & 7));
        dst[r * ld + at] = v;
    }
}

// Row images: dst[cidx * ldt + kvpos(r)] = bf16(src[r * cols + cidx]) for r < rows, 0 for rows <= r < rows_pad; rows of
// dst beyond `cols` (up to cols_pad) are zero.  32x32 tiles through LDS so both sides coalesce (kvpos permutes inside
// 16-element groups, ldt is a multiple of 32).
// Tiled images (TILED; cols_pad a multiple of 512): key tile T = r / 32 of the image at `dst`, column chunk C = cidx / 512:
// block (T * cols_pad / 512 + C) of 512 x 32 elements; in it row cidx % 512 is 64 bytes = the tile's 32 key positions
// (kvpos order), 16-byte chunk q stored at q ^ ((row >> 2) & 3) -- the byte order of the tandem kernel's LDS Vt buffer.
__device__ __forceinline__ unsigned short to_bf16_elem(double x) { return (unsigned short)f32_to_bf16_rne(__double2float_rn(x)); }
__device__ __forceinline__ unsigned short to_bf16_elem(unsigned short x) { return x; }      // rounded by the host already

template <typename SRC, bool TILED>
__global__ void cvt_d2bf_t_kernel(const SRC *__restrict__ src, unsigned short *__restrict__ dst,
                                  long rows, long rows_pad, int cols, int cols_pad, long ldt) {
    __shared__ unsigned short tile[32][33];
    const long r0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 256 threads: ty in 0..7
    for (int k = ty; k < 32; k += 8) {
        const long r = r0 + k;
        const int cc = c0 + tx;
        tile[k][tx] = (r < rows && cc < cols) ? to_bf16_elem(cvt_src_load(&src[r * cols + cc])) : (unsigned short)0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int cc = c0 + k;
        const long r = r0 + tx;
        if (cc >= cols_pad || r >= rows_pad) continue;
        if constexpr (TILED) {
            const int row = cc & 511, pos = (int)bf16_kvpos(tx);
            const size_t block = (size_t)blockIdx.x * (cols_pad >> 9) + (cc >> 9);
            dst[block * (512 * 32) + (size_t)row * 32 + ((((pos >> 3) ^ ((row >> 2) & 3)) << 3) | (pos & 7))] = tile[tx][k];
        } else {
            dst[(size_t)cc * ldt + r0 + bf16_kvpos(tx)] = tile[tx][k];
        }
    }
}

// ---------------------------------------------------------------------------
// host-side launch logic
// ---------------------------------------------------------------------------
int bf16_pad_dk(int dk) { return dk <= 64 ? 64 : dk <= 128 ? 128 : dk <= 256 ? 256 : 512; }
int bf16_chunk_dv(int dv) { return dv <= 64 ? 64 : dv <= 128 ? 128 : dv <= 256 ? 256 : 512; }
int bf16_pad_dv(int dv) { const int ch = bf16_chunk_dv(dv); return (dv + ch - 1) / ch * ch; }
long bf16_pad_n(long n) { return (n + 31) / 32 * 32; }

// Which shapes take the two-query-blocks-per-wave kernel: dk and dv both within one 256-wide
// operand ($SDPA_DEBUG bf16_duo=0 keeps them on the general kernel -- both are exact paths; the switch
// exists for A/B timing).
bool bf16_uses_duo(int dk, int dv) {
    static const int enabled = sdpa_debug_int("bf16_duo", 1);
    return enabled && dk <= 256 && dv <= 256;
}
// kernels with a fixed reference exponent flag blocks for a second pass: one int per (split, 128-row block)
bool bf16_needs_redo(int dk, int dv) { return bf16_chunk_dv(dv) == 512 || bf16_uses_duo(dk, dv); }

// workspace: [kv_splits x m x ws_ld] contrib, [kv_splits x m] lmax, [kv_splits x m] lsum when the
// shard is split, then (dv > 256) one redo flag per (split, q block) for the wide kernel
size_t bf16_workspace_bytes(int m, int n_local, int dk, int dv) {
    if (m <= 0) return 0;
    const int s = pick_kv_splits_bf16(m, n_local, dk, dv);
    const int ws_ld = (dv + 3) / 4 * 4;
    size_t bytes = s <= 1 ? 0 : (size_t)s * m * ((size_t)ws_ld + 2) * sizeof(float);
    if (bf16_needs_redo(dk, dv)) bytes += (size_t)((m + kQRowsPerBlock - 1) / kQRowsPerBlock) * s * sizeof(int);
    return bytes;
}
void bf16_carve_workspace(Bf16Args &a, void *ws, int ws_ld) {
    char *p = static_cast<char *>(ws);
    if (a.kv_splits > 1) {
        a.ws_ld = ws_ld;
        a.ws_contrib = reinterpret_cast<float *>(p);
        a.ws_lmax = a.ws_contrib + (size_t)a.kv_splits * a.m * a.ws_ld;
        a.ws_lsum = a.ws_lmax + (size_t)a.kv_splits * a.m;
        p = reinterpret_cast<char *>(a.ws_lsum + (size_t)a.kv_splits * a.m);
    }
    a.redo = bf16_needs_redo(a.dk, a.dv) ? reinterpret_cast<int *>(p) : nullptr;
}

int pick_kv_splits_bf16(int m, int n_local, int dk, int dv) {
    if (m <= 0 || n_local <= 0) return 1;
    if (bf16_uses_duo(dk, dv)) {            // 256-row workgroups, one per CU
        const int nqb = (m + kDuoRows - 1) / kDuoRows;
        const int ntiles = (n_local + kKvTile - 1) / kKvTile;
        int want = (256 + nqb - 1) / nqb, cap = ntiles / 8;
        if (cap < 1) cap = 1;
        if (want > cap) want = cap;
        if (want > 64) want = 64;
        if (want < 1) want = 1;
        const double kernel_s = 2.0 * m * (double)n_local * (dk + dv) / 1.2e15;
        const double slab_s = 2.0 * m * (double)((dv + 3) / 4 * 4) * sizeof(float) / 3.0e12;
        return splits_for_full_rounds(nqb, 256, want, cap, kernel_s, slab_s);      // a fuller last round (sdpa_internal.h)
    }
    const int nqb = (m + kQRowsPerBlock - 1) / kQRowsPerBlock;
    const int chunks = bf16_pad_dv(dv) / bf16_chunk_dv(dv);
    const int ntiles = (n_local + kKvTile - 1) / kKvTile;
    const int per_cu = (bf16_pad_dk(dk) + 2 * bf16_chunk_dv(dv) > 512) ? 1 : 2;
    int want = (256 * per_cu + nqb * chunks - 1) / (nqb * chunks);
    int cap = ntiles / 8;
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want > 64) want = 64;
    if (want < 1) want = 1;
    const double kernel_s = 2.0 * m * (double)n_local * (dk + dv) / 1.1e15;
    const double slab_s = 2.0 * m * (double)((dv + 3) / 4 * 4) * sizeof(float) / 3.0e12;
    return splits_for_full_rounds((long)nqb * chunks, 256 * per_cu, want, cap, kernel_s, slab_s);
}

template <int DK, int DVC, int ABL = 0>
static hipError_t launch_bf16_pipe(const Bf16Args &a, hipStream_t s) {
    const int nqb = (a.m + kQRowsPerBlock - 1) / kQRowsPerBlock;
    const int chunks = bf16_pad_dv(a.dv) / DVC;
    const int ntiles = (a.n_local + kKvTile - 1) / kKvTile;
    const int tiles_per_split = (ntiles + a.kv_splits - 1) / a.kv_splits;
    const int kv_per_split = tiles_per_split > 0 ? tiles_per_split * kKvTile : kKvTile;
    const size_t lds = (size_t)2 * (kKvTile * DK + DVC * 36) * sizeof(unsigned short);
    static std::atomic<bool> attr_done[64];   // (zero-initialised; set from any enqueue thread)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_bf16_pipe_kernel<DK, DVC, ABL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[dev] = true;
    }
    const float scale = 1.0f;   // attention-mpi.c:208's 1/sqrt(dk) (and log2 e) live in the Q image
    hipLaunchKernelGGL((fused_bf16_pipe_kernel<DK, DVC, ABL>), dim3(nqb * chunks * a.kv_splits), dim3(256), lds,
                       s, a, kv_per_split, nqb, chunks, scale);
    note_launch("fused_bf16_pipe_kernel", 3, DK, DVC, ABL, 0, 0, nqb * chunks * a.kv_splits, a.kv_splits, 0, a.m, a.n_local);
    return hipGetLastError();
}

template <int DK>
static hipError_t launch_bf16_tandem(const Bf16Args &a, hipStream_t s) {
    const int nqb = (a.m + kQRowsPerBlock - 1) / kQRowsPerBlock;
    const int chunks = bf16_pad_dv(a.dv) / 512;
    const int ntiles = (a.n_local + kKvTile - 1) / kKvTile;
    const int tiles_per_split = (ntiles + a.kv_splits - 1) / a.kv_splits;
    const int kv_per_split = tiles_per_split > 0 ? tiles_per_split * kKvTile : kKvTile;
    const size_t lds = ((size_t)2 * kKvTile * DK + (size_t)2 * 512 * kKvTile) * sizeof(unsigned short) + 16384;
    static std::atomic<bool> attr_done[64];   // (zero-initialised; set from any enqueue thread)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_bf16_tandem_kernel<DK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((fused_bf16_tandem_kernel<DK>), dim3(nqb * chunks * a.kv_splits), dim3(256), lds, s,
                       a, kv_per_split, nqb, chunks, 1.0f);
    note_launch("fused_bf16_tandem_kernel", 1, DK, 0, 0, 0, 0, nqb * chunks * a.kv_splits, a.kv_splits, 0, a.m, a.n_local);
    return hipGetLastError();
}

// the persistent form (fused_bf16_tandem_stream_kernel): the same grid, the same arguments + where its ready words live
template <int DK>
static hipError_t launch_bf16_tandem_streamed(const Bf16Args &a, const StreamArgs &st, hipStream_t s) {
    const int nqb = (a.m + kQRowsPerBlock - 1) / kQRowsPerBlock;
    const int chunks = bf16_pad_dv(a.dv) / 512;
    const int ntiles = (a.n_local + kKvTile - 1) / kKvTile;
    const int tiles_per_split = (ntiles + a.kv_splits - 1) / a.kv_splits;
    const int kv_per_split = tiles_per_split > 0 ? tiles_per_split * kKvTile : kKvTile;
    const size_t lds = ((size_t)2 * kKvTile * DK + (size_t)2 * 512 * kKvTile) * sizeof(unsigned short) + 16384;
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_bf16_tandem_stream_kernel<DK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((fused_bf16_tandem_stream_kernel<DK>), dim3(nqb * chunks * a.kv_splits), dim3(256), lds, s,
                       a, kv_per_split, nqb, chunks, 1.0f, st);
    note_launch("fused_bf16_tandem_stream_kernel", 1, DK, 0, 0, 0, 0, nqb * chunks * a.kv_splits, a.kv_splits, 0, a.m, a.n_local);
    return hipGetLastError();
}

template <int DK, int DV>
static hipError_t launch_bf16_duo(const Bf16Args &a, hipStream_t s) {
    const int nqb = (a.m + kDuoRows - 1) / kDuoRows;
    const int nqb128 = (a.m + kQRowsPerBlock - 1) / kQRowsPerBlock;
    const int ntiles = (a.n_local + kKvTile - 1) / kKvTile;
    const int tiles_per_split = (ntiles + a.kv_splits - 1) / a.kv_splits;
    const int kv_per_split = tiles_per_split > 0 ? tiles_per_split * kKvTile : kKvTile;
    const size_t lds = DuoCfg<DK, DV>::lds_bytes;
    static std::atomic<bool> attr_done[64];   // (zero-initialised; set from any enqueue thread)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_bf16_duo_kernel<DK, DV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[dev] = true;
    }
    const float scale = 1.0f;   // attention-mpi.c:208's 1/sqrt(dk) (and log2 e) live in the Q image
    hipLaunchKernelGGL((fused_bf16_duo_kernel<DK, DV>), dim3(nqb * a.kv_splits), dim3(256), lds, s, a,
                       kv_per_split, nqb, nqb128, scale);
    note_launch("fused_bf16_duo_kernel", 2, DK, DV, 0, 0, 0, nqb * a.kv_splits, a.kv_splits, 0, a.m, a.n_local);
    return hipGetLastError();
}

static hipError_t launch_shard_partial_bf16_impl(const Bf16Args &args, const StreamArgs *st, hipStream_t s);

hipError_t launch_shard_partial_bf16(const Bf16Args &args, hipStream_t s) {
    return launch_shard_partial_bf16_impl(args, nullptr, s);
}

// dims whose main kernel has a persistent form: the tandem kernel's (value columns in 512-wide chunks, i.e. dv > 256)
bool bf16_stream_launch_supported(int dk, int dv) {
    return dk >= 1 && dk <= 512 && bf16_tiled(dv);
}

hipError_t launch_shard_partial_bf16_streamed(const Bf16Args &a, const StreamArgs &st, hipStream_t s) {
    if (!bf16_stream_launch_supported(a.dk, a.dv) || !st.flags || !st.status || !st.abort || st.n_chunks < 1 ||
        st.n_chunks > kStreamMaxChunks || a.n_local <= 0)
        return hipErrorInvalidValue;
    return launch_shard_partial_bf16_impl(a, &st, s);
}

static hipError_t launch_shard_partial_bf16_impl(const Bf16Args &args, const StreamArgs *st, hipStream_t s) {
    static std::atomic<int> generation{0x5d9a0000};
    Bf16Args a = args;
    if (a.ws_rows <= 0) a.ws_rows = a.m;
    a.redo_gen = generation.fetch_add(1) + 1;
    const int kp = bf16_pad_dk(a.dk), vc = bf16_chunk_dv(a.dv);
    if (a.dk > 512 || a.ldq != kp || a.ldk != kp) return hipErrorInvalidValue;
    hipError_t e = hipErrorInvalidValue;
    LaunchNote main_note = {};
    bool have_main = false;
#ifdef SDPA_ABLATIONS   // tools/ builds only: the shipped library never reads $SDPA_DEBUG tune
    static const int tune = sdpa_debug_int("tune", 0);
    if (kp == 512 && vc == 256 && ((tune >> 8) & 15)) {   // timing-only ablations, pipelined kernel
        switch ((tune >> 8) & 15) {
            case 1: return launch_bf16_pipe<512, 256, 1>(a, s);     // no DMA / V staging
            case 2: return launch_bf16_pipe<512, 256, 2>(a, s);     // no LDS fragment reads
            case 8: return launch_bf16_pipe<512, 256, 8>(a, s);     // no barrier (racy)
            case 9: return launch_bf16_pipe<512, 256, 9>(a, s);     // no staging, no barrier
            default: return launch_bf16_pipe<512, 256, 11>(a, s);   // MFMA + softmax only
        }
    }
#endif
    // every byte offset inside the Vt image is carried in 32 bits by the staging code
    if ((size_t)bf16_pad_dv(a.dv) * (size_t)bf16_pad_n(a.n_local) * 2u > 0xffffffffull) return hipErrorInvalidValue;
    a.tiled = vc == 512 ? 1 : 0;                // (the redo pass reads the same images: it has to know their layout)
    if (vc == 512) {
        if ((reinterpret_cast<uintptr_t>(a.K) & 15) || (reinterpret_cast<uintptr_t>(a.Vt) & 15) || !a.redo)
            return hipErrorInvalidValue;
#ifdef SDPA_ABLATIONS   // tools/ builds only ($SDPA_DEBUG tune bit 12): the STREAM kernel on resident images, every ready word raised beforehand --
        // what the persistent form's waits and extra arguments cost the kernel itself (profiles/r05/bf16_stream_kernel_resident_ab.log)
        static StreamArgs self_st = {};
        if (!st && (tune & 4096)) {
            if (!self_st.flags) {
                unsigned *w = nullptr;
                int *stat = nullptr;
                const size_t words = (size_t)(kStreamMaxChunks + kStreamMaxPieces) * kStreamFlagStride;
                if (hipMalloc((void **)&w, words * sizeof(unsigned)) != hipSuccess || hipMalloc((void **)&stat, 64) != hipSuccess) return hipErrorOutOfMemory;
                (void)hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(w), 0x5eed0001u, words);
                (void)hipMemset(stat, 0, 64);
                self_st.flags = w; self_st.gen = 0x5eed0001u; self_st.status = stat; self_st.abort = reinterpret_cast<unsigned *>(stat) + 4;
                self_st.timeout_ticks = 100000000ull;
            }
            const int ntiles_ = (a.n_local + kKvTile - 1) / kKvTile;
            self_st.n_chunks = 1;
            self_st.chunk_end[0] = (ntiles_ + a.kv_splits - 1) / a.kv_splits;
            self_st.q_piece_blocks = 64;
            st = &self_st;
        }
#endif
        if (st) {
            switch (kp) {
                case 64: e = launch_bf16_tandem_streamed<64>(a, *st, s); break;
                case 128: e = launch_bf16_tandem_streamed<128>(a, *st, s); break;
                case 256: e = launch_bf16_tandem_streamed<256>(a, *st, s); break;
                default: e = launch_bf16_tandem_streamed<512>(a, *st, s); break;
            }
        } else {
            switch (kp) {
                case 64: e = launch_bf16_tandem<64>(a, s); break;
                case 128: e = launch_bf16_tandem<128>(a, s); break;
                case 256: e = launch_bf16_tandem<256>(a, s); break;
                default: e = launch_bf16_tandem<512>(a, s); break;
            }
        }
        if (e != hipSuccess) return e;
        main_note = last_launch_note();
        have_main = true;
        // general kernel (256-column chunks, in-loop rescale) over the blocks the tandem one flagged
        switch (kp) {
            case 64: e = launch_bf16_pipe<64, 256>(a, s); break;
            case 128: e = launch_bf16_pipe<128, 256>(a, s); break;
            case 256: e = launch_bf16_pipe<256, 256>(a, s); break;
            default: e = launch_bf16_pipe<512, 256>(a, s); break;
        }
    }
    if (reinterpret_cast<uintptr_t>(a.K) & 15) return hipErrorInvalidValue;   // LDS-DMA moves 16-byte chunks
    if (bf16_uses_duo(a.dk, a.dv)) {
        if ((reinterpret_cast<uintptr_t>(a.Vt) & 15) || !a.redo) return hipErrorInvalidValue;
#define SDPA_DCASE(KP, VC) if (kp == KP && vc == VC) e = launch_bf16_duo<KP, VC>(a, s);
        SDPA_DCASE(64, 64)  SDPA_DCASE(64, 128)  SDPA_DCASE(64, 256)
        SDPA_DCASE(128, 64) SDPA_DCASE(128, 128) SDPA_DCASE(128, 256)
        SDPA_DCASE(256, 64) SDPA_DCASE(256, 128) SDPA_DCASE(256, 256)
#undef SDPA_DCASE
        if (e != hipSuccess) return e;
        main_note = last_launch_note();
        have_main = true;
        // then the general kernel over the blocks the duo kernel flagged (in-loop rescale)
    }
#define SDPA_BCASE(KP, VC) \
    if (kp == KP && vc == VC) e = launch_bf16_pipe<KP, VC>(a, s);
    SDPA_BCASE(64, 64)  SDPA_BCASE(64, 128)  SDPA_BCASE(64, 256)
    SDPA_BCASE(128, 64) SDPA_BCASE(128, 128) SDPA_BCASE(128, 256)
    SDPA_BCASE(256, 64) SDPA_BCASE(256, 128) SDPA_BCASE(256, 256)
    SDPA_BCASE(512, 64) SDPA_BCASE(512, 128) SDPA_BCASE(512, 256)
#undef SDPA_BCASE
    if (e != hipSuccess) return e;
    if (have_main) set_launch_note(main_note);       // the launch's kernel is the main one, not its redo pass
    if (a.kv_splits > 1 && !a.defer_merge) {
        PartialArgs p = {};
        p.contrib = a.contrib; p.ldo = a.ldo; p.lmax = a.lmax; p.lsum = a.lsum;
        p.m = a.m; p.dv = a.dv; p.kv_splits = a.kv_splits;
        p.ws_contrib = a.ws_contrib; p.ws_ld = a.ws_ld; p.ws_lmax = a.ws_lmax; p.ws_lsum = a.ws_lsum;
        p.ws_rows = a.ws_rows;
        e = launch_split_merge(p, s);
    }
    return e;
}

static hipError_t launch_cvt_rows(const double *src, unsigned short *dst, long rows, long rows_pad, int cols, int ld, double mult,
                                  int swz, hipStream_t s) {
    if (rows_pad <= 0) return hipSuccess;
    long g = (rows_pad * ld + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(cvt_d2bf_kernel, dim3((unsigned)g), dim3(256), 0, s, src, dst, rows, rows_pad, cols, ld, mult, swz);
    return hipGetLastError();
}

hipError_t launch_cvt_d2bf(const double *src, unsigned short *dst, long rows, int cols, int ld,
                           hipStream_t s) {
    return launch_cvt_rows(src, dst, rows, rows, cols, ld, 1.0, 0, s);
}

// the K image of a (dk, dv) shape: rows [rows, rows_pad) zero; tiled images (dv > 256) with the row's chunk swizzle
hipError_t launch_cvt_d2bf_k(const double *src, unsigned short *dst, long rows, long rows_pad, int dk, int dv, hipStream_t s) {
    return launch_cvt_rows(src, dst, rows, rows_pad, dk, bf16_pad_dk(dk), 1.0, bf16_k_swz(dk, dv), s);
}

// the Q image of the bf16 kernels: bf16(Q * log2(e)/sqrtf(dk)), ONE rounding from fp64 -- the softmax
// scale (attention-mpi.c:208) and the change of base for v_exp_f32 folded into the operand
hipError_t launch_cvt_d2bf_q(const double *src, unsigned short *dst, long rows, int dk, int ld, hipStream_t s) {
    const float c = 1.44269504088896340736f * (1.0f / sqrtf((float)dk));
    return launch_cvt_rows(src, dst, rows, rows, dk, ld, (double)c, 0, s);
}

hipError_t launch_cvt_d2bf_t_part(const double *src, unsigned short *dst, long rows, long rows_pad, int cols,
                                  int cols_pad, long ldt, hipStream_t s) {
    if (rows_pad <= 0 || cols_pad <= 0) return hipSuccess;
    const unsigned gx = (unsigned)((rows_pad + 31) / 32), gy = (unsigned)((cols_pad + 31) / 32);
    if (bf16_tiled(cols))
        hipLaunchKernelGGL((cvt_d2bf_t_kernel<double, true>), dim3(gx, gy), dim3(256), 0, s, src, dst, rows, (long)gx * 32, cols,
                           cols_pad, ldt);
    else
        hipLaunchKernelGGL((cvt_d2bf_t_kernel<double, false>), dim3(gx, gy), dim3(256), 0, s, src, dst, rows, rows_pad, cols,
                           cols_pad, ldt);
    return hipGetLastError();
}

hipError_t launch_cvt_bf_t_part(const unsigned short *src, unsigned short *dst, long rows, long rows_pad, int cols,
                                int cols_pad, long ldt, hipStream_t s) {
    if (rows_pad <= 0 || cols_pad <= 0) return hipSuccess;
    const unsigned gx = (unsigned)((rows_pad + 31) / 32), gy = (unsigned)((cols_pad + 31) / 32);
    if (bf16_tiled(cols))
        hipLaunchKernelGGL((cvt_d2bf_t_kernel<unsigned short, true>), dim3(gx, gy), dim3(256), 0, s, src, dst, rows, (long)gx * 32,
                           cols, cols_pad, ldt);
    else
        hipLaunchKernelGGL((cvt_d2bf_t_kernel<unsigned short, false>), dim3(gx, gy), dim3(256), 0, s, src, dst, rows, rows_pad,
                           cols, cols_pad, ldt);
    return hipGetLastError();
}

hipError_t launch_cvt_d2bf_t(const double *src, unsigned short *dst, long rows, int cols,
                             int cols_pad, long ldt, hipStream_t s) {
    return launch_cvt_d2bf_t_part(src, dst, rows, ldt, cols, cols_pad, ldt, s);
}

void dma_audit_read_bf16(unsigned long long out[2]) {
    out[0] = out[1] = 0;
#ifdef SDPA_DMA_ASSERT
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bf16_audit), 2 * sizeof(unsigned long long));
#endif
}

// (sdpa_internal.h: preload_kernels_*) touching one kernel makes the runtime load this translation unit's code object for the
// current device NOW -- not in front of the first launch that needs it, possibly behind a resident persistent launch
hipError_t preload_kernels_bf16() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&cvt_d2bf_kernel));
}

}  // namespace sdpa
