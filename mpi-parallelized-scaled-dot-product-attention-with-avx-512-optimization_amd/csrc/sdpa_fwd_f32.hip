// sdpa_fwd_f32.hip -- the fused online-softmax attention kernel for gfx950 (CDNA4).
//
// Replaces the reference's per-row hot loop (paths relative to the reference tree):
//   online_softmax_attention  attention-mpi.c:168-189
//   dot_avx512                :103-121   -> QK^T on v_mfma_f32_32x32x2_f32
//   axpy_avx512               :123-140   -> P.V  on v_mfma_f32_32x32x2_f32
//   memset_zero_scale         :142-166   -> accumulator rescale, only when a tile
//                                           moves the running max of some row
// for ALL query rows of a batch against one K/V shard, producing the same
// shard-local triple the reference produces per row: un-normalised contrib[dv],
// lmax, lsum.
//
// Design (MI355X first; nothing here is a translation of the AVX-512 code):
//   * workgroup = 4 waves, each wave owns 32 query rows; a workgroup walks its
//     K/V range in 32-row tiles staged through LDS (double buffered, one barrier
//     per tile, next tile's global loads in flight under the current tile's MFMAs).
//   * swapped product S^T = K.Q^T: the MFMA D layout then gives every lane ONE
//     query row (lane&31) and 16 of the tile's 32 key rows, so row max / row sum
//     are lane-local apart from one exchange with lane^32.
//   * P^T in the D layout is directly the B operand of O^T += V^T.P^T when the
//     two k-slots of step r are key rows crow(r,0) / crow(r,1): no LDS round trip
//     and no shuffle between the two contractions.
//   * Q (32 x dk per wave) lives in registers for the whole K/V walk, O^T in
//     4 x 16 accumulator registers.
//   * f32-input MFMA is an exact fmaf chain (1/16 of the bf16 MFMA rate =
//     157.3 TFLOP/s peak): "fp32 compute" as the reference, same rounding class.
//   * in-GPU K/V splits (flash-decoding style) fill the 256 CUs when there are
//     few query blocks; the split merge is the reference's own (max,sum,contrib)
//     merge algebra (:340-362) applied inside one GPU.
//
// Kernels in this file:
//   fused_pipelined_kernel<DK,DV>   dense dk, dv in {64,128}: LDS-DMA staging, XOR-swizzled K image,
//                                   two live score tiles (the shipped path of every BASELINE fp32
//                                   config; 143 TFLOP/s = 91 % of peak at the metric shape); and dense
//                                   {256,256}, {256,128}, {128,256} at one wave per SIMD (O^T in the
//                                   accumulator file, asm score chains: 137 TFLOP/s at dk = dv = 256)
//   fused_partial_kernel<DKP,DVP>   any dk <= 256 (padded to 32/64/128/256), any dv (chunks of <= 128
//                                   columns): register-staged
//   (fused_dksplit_pipe_kernel<DKS,DVS,QB>, 256 < dk <= 1024 and non-dense 128 < dk <= 256 with dv > 128, lives in
//    sdpa_fwd_f32_dksplit.hip)
//   generic_partial_kernel          dk > 1024: VALU-only correctness path
//   split_merge_kernel              merge of the in-GPU K/V splits
//
#include "sdpa_f32_device.h"
#include "sdpa_debug.h"
#include <mutex>
#include <vector>

#include <math.h>
#include <algorithm>
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <stdint.h>
#include <type_traits>

// -DSDPA_RANGE_REDO=0 builds the fp32 pipelined kernels without their range check and second pass
// (tools/build_variant.sh: an A/B of what the check costs; never the shipped library)
#ifndef SDPA_RANGE_REDO
#define SDPA_RANGE_REDO 1
#endif

namespace sdpa {

// dv > 128 is processed in chunks of 128 columns by separate workgroups (n_chunks > 1; the score
// tile is recomputed per chunk); dk up to 256 keeps the whole Q fragment in registers (128 VGPRs,
// one wave per SIMD then).
template <int DKP, int DVP>
__global__ __launch_bounds__(256, (DKP > 128) ? 1 : 2) void fused_partial_kernel(
    PartialArgs a, int kv_per_split, int n_qblocks, int n_chunks, float scale) {
    constexpr int NU = DKP / 8;     // 16-byte K reads (4 MFMA k-steps each) per tile per lane
    constexpr int NT = DVP / 32;    // 32-column O^T tiles; also floats per V read
    constexpr int KLD = DKP + 4;    // padded K row (floats): ds_read_b128 column reads conflict-free
    constexpr int KPT = DKP / 32;   // float4 staged per thread per K tile
    constexpr int VPT = DVP / 32;
    constexpr int KTILE = kKvTile * KLD;
    constexpr int VTILE = kKvTile * DVP;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *const Ks = smem;                 // [2][KTILE]
    float *const Vs = smem + 2 * KTILE;     // [2][VTILE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;               // MFMA row/col index of this lane
    const int hi = lane >> 5;               // which k-slot / accumulator half

    int work = xcd_remap(blockIdx.x, gridDim.x);
    const int qblock = work % n_qblocks;
    work /= n_qblocks;
    const int chunk = work % n_chunks;
    const int split = work / n_chunks;
    const int qrow = qblock * kQRowsPerBlock + wave * 32 + li;   // this lane's query row
    const int dv0 = chunk * DVP;                                  // first V / output column

    const int kv_begin = split * kv_per_split;
    const int kv_end = min(a.n_local, kv_begin + kv_per_split);
    const int ntiles = kv_end > kv_begin ? (kv_end - kv_begin + kKvTile - 1) / kKvTile : 0;

    const float c = scale * 1.44269504088896340736f;   // scores -> log2 domain

    // ---- Q fragment: B operand of S^T = K.Q^T.  k-slot mapping (shared with the K
    //      reads below): MFMA step (u,e) of half-wave hi uses dk index 8u + 4hi + e.
    float4 qf[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int col = 8 * u + 4 * hi;
        qf[u] = (qrow < a.m && col < a.ldq)
                    ? *reinterpret_cast<const float4 *>(a.Q + (size_t)qrow * a.ldq + col)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    f32x16 oacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float m_run = -INFINITY;   // running max of raw dots (scale > 0 keeps the order)
    float l_run = 0.f;         // this half-wave's share of the running sum

    // Staging loads are UNCONDITIONAL (no exec-masked branches, so the loads stay in flight
    // under the MFMAs): rows past the shard end are clamped to its last row (their scores are
    // masked to -inf below, and 0 * finite = 0 in P.V), columns past the leading dimension are
    // clamped to the last in-row float4 (the matching Q columns are zero; V columns past dv
    // are never stored).
    f32x4 kreg[KPT], vreg[VPT];   // native vectors: plain SSA values after unrolling
    const int ldk_last = a.ldk - 4, ldv_last = a.ldv - 4;
    unsigned koff[KPT], voff[VPT];           // loop-invariant per-lane byte offsets (full tiles)
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int idx = tid + 256 * i;
        koff[i] = (unsigned)((idx / (DKP / 4)) * a.ldk + min(4 * (idx % (DKP / 4)), ldk_last)) * 4u;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = tid + 256 * i;
        voff[i] = (unsigned)((idx / (DVP / 4)) * a.ldv + min(dv0 + 4 * (idx % (DVP / 4)), ldv_last)) * 4u;
    }
    auto tile_gload = [&](int tile) __attribute__((always_inline)) {
        const int base = kv_begin + tile * kKvTile;          // wave-uniform
        const int last = kv_end - 1 - base;                  // last valid row of this tile
        // scalar 64-bit base + unsigned 32-bit per-lane byte offset (saddr + voffset form)
        const char *kb = reinterpret_cast<const char *>(a.K + (size_t)base * a.ldk);
        const char *vb = reinterpret_cast<const char *>(a.V + (size_t)base * a.ldv);
        if (last >= kKvTile - 1) {                           // full tile: nothing to clamp
#pragma unroll
            for (int i = 0; i < KPT; ++i) kreg[i] = *reinterpret_cast<const f32x4 *>(kb + koff[i]);
#pragma unroll
            for (int i = 0; i < VPT; ++i) vreg[i] = *reinterpret_cast<const f32x4 *>(vb + voff[i]);
        } else {                                             // ragged last tile: clamp the rows
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const int idx = tid + 256 * i;
                const int row = min(idx / (DKP / 4), last);
                const int col = min(4 * (idx % (DKP / 4)), ldk_last);
                kreg[i] = *reinterpret_cast<const f32x4 *>(kb + (unsigned)(row * a.ldk + col) * 4u);
            }
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int idx = tid + 256 * i;
                const int row = min(idx / (DVP / 4), last);
                const int col = min(dv0 + 4 * (idx % (DVP / 4)), ldv_last);
                vreg[i] = *reinterpret_cast<const f32x4 *>(vb + (unsigned)(row * a.ldv + col) * 4u);
            }
        }
    };
    auto tile_lstore = [&](int buf) __attribute__((always_inline)) {
        float *kd = Ks + buf * KTILE;
        float *vd = Vs + buf * VTILE;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / (DKP / 4), c4 = idx % (DKP / 4);
            *reinterpret_cast<f32x4 *>(kd + row * KLD + 4 * c4) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / (DVP / 4), c4 = idx % (DVP / 4);
            *reinterpret_cast<f32x4 *>(vd + row * DVP + 4 * c4) = vreg[i];
        }
    };

    if (ntiles > 0) {
        tile_gload(0);
        tile_lstore(0);
    }
    __syncthreads();


    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        // next tile's global loads, in flight under this tile's MFMAs.  Unconditional (the
        // last iteration re-fetches its own tile into the idle buffer) so that the staged
        // registers stay plain SSA values -- no control flow, no scratch.
        tile_gload(min(t + 1, ntiles - 1));

        // ---- S^T tile = K_tile . Q^T   (A = K rows from LDS, B = Q from registers)
        const float *kt = Ks + cur * KTILE + li * KLD + 4 * hi;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        // K fragments are read two 16-byte pieces (8 MFMAs = 512 cycles) ahead of their use
        float4 kf0 = *reinterpret_cast<const float4 *>(kt);
        float4 kf1 = *reinterpret_cast<const float4 *>(kt + 8);
#pragma unroll
        for (int u = 0; u < NU; u += 2) {
            float4 kn0 = kf0, kn1 = kf1;
            if (u + 2 < NU) {
                kn0 = *reinterpret_cast<const float4 *>(kt + 8 * (u + 2));
                kn1 = *reinterpret_cast<const float4 *>(kt + 8 * (u + 3));
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the reads AHEAD of this step's MFMAs
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf0.x, qf[u].x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf0.y, qf[u].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf0.z, qf[u].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf0.w, qf[u].w, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf1.x, qf[u + 1].x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf1.y, qf[u + 1].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf1.z, qf[u + 1].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf1.w, qf[u + 1].w, s, 0, 0, 0);
            kf0 = kn0;
            kf1 = kn1;
        }

        // ragged last tile: key rows past the shard end contribute exp(-inf) = 0
        const int valid = kv_end - (kv_begin + t * kKvTile);
        if (valid < kKvTile) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (crow(r, hi) >= valid) s[r] = -INFINITY;
        }

        // ---- online softmax, one query row per lane pair (lane, lane^32)
        float tmax = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        if (__any(m_new > m_run)) {           // wave-uniform: rare after the first tiles
            const float alpha = fast_exp2((m_run - m_new) * c);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[tt][r] *= alpha;
            l_run *= alpha;
            m_run = m_new;
        }
        const float mc = m_run * c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fast_exp2(fmaf(s[r], c, -mc));
            l_run += s[r];
        }

        // ---- O^T += V_tile^T . P^T   (A = V columns from LDS, B = P from registers)
        // V fragments are read two steps (8 MFMAs) ahead; the next tile's staged registers go
        // to the other LDS buffer half-way through, under the MFMAs.
        const float *vt = Vs + cur * VTILE + NT * li + 4 * hi * DVP;   // crow(r,hi) = crow(r,0) + 4hi
        VFrag<NT> vf0 = VFrag<NT>::load(vt + crow(0, 0) * DVP);
        VFrag<NT> vf1 = VFrag<NT>::load(vt + crow(1, 0) * DVP);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            VFrag<NT> vn0 = vf0, vn1 = vf1;
            if (r + 2 < 16) {
                vn0 = VFrag<NT>::load(vt + crow(r + 2, 0) * DVP);
                vn1 = VFrag<NT>::load(vt + crow(r + 3, 0) * DVP);
            }
            if (r == 8) tile_lstore(cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
                oacc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf0.v[tt], s[r], oacc[tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
                oacc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf1.v[tt], s[r + 1], oacc[tt], 0, 0, 0);
            vf0 = vn0;
            vf1 = vn1;
        }
        __syncthreads();
    }

    // ---- epilogue: the shard-local triple of attention-mpi.c:188 for this row
    const float l_tot = l_run + __shfl_xor(l_run, 32);
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
        for (int r = 0; r < 16; ++r) {
            const int col0 = NT * crow(r, hi);
            if constexpr (NT == 4) {
                if (dv0 + col0 + 3 < a.dv) {
                    *reinterpret_cast<float4 *>(orow + col0) =
                        make_float4(oacc[0][r], oacc[1][r], oacc[2][r], oacc[3][r]);
                    continue;
                }
            }
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
                if (dv0 + col0 + tt < a.dv) orow[col0 + tt] = oacc[tt][r];
        }
        if (hi == 0 && chunk == 0) {
            omax[qrow] = m_run * scale;
            osum[qrow] = l_tot;
        }
    }
}

// ---------------------------------------------------------------------------
// Software-pipelined variant for dense dk, dv in {64, 128, 256} (leading dimensions equal to
// the dims, i.e. no padding columns).  Same maths and same outputs as
// fused_partial_kernel; what changes is the schedule inside a wave:
//   * K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
//     instruction, no staging registers).  The DMA destination is lane-linear, so the
//     conflict-free K image is an XOR swizzle of the 16-byte chunks (chunk c of row r sits
//     at position c ^ (r & 15)) applied on the per-lane SOURCE address and undone by the
//     reads; V rows are read whole and need none.
//   * two score tiles are live: while the matrix pipe runs S^T(t+1) = K(t+1).Q^T, the VALU
//     turns S^T(t) into P(t) (fma + exp2 + row-sum), 2 values per 8 MFMAs; while it runs
//     O^T += V(t)^T.P(t)^T, the VALU reduces the row max of S^T(t+1).  The only serial
//     pieces left per tile are the (rare) accumulator rescale and the barrier.
//   * K is staged two tiles ahead, V one tile ahead, in two buffers each.
// ---------------------------------------------------------------------------
// ABL: timing-only ablation switches (results are wrong when non-zero; $SDPA_DEBUG tune selects them):
//   1 = no DMA / no barrier in the steady state, 2 = no LDS fragment reads, 4 = no softmax VALU
// MERGE: 1 = the launch merges its K/V splits itself (arrival words, $SDPA_DEBUG split_merge=kernel); the shipped
// default instantiation carries none of that code
// SK: 1 = stream-K work distribution (round 4).  The launch's n_qblocks x ntiles tile steps (query block
// major) are cut into gridDim.x equal runs of `kv_per_split` steps, one per workgroup = one per RESIDENT
// workgroup slot of the stream's compute units; a workgroup walks its run piece by piece (a piece = the part
// of one query block's K/V range inside the run) and writes each piece's partial triple into slab
// (workgroup - first workgroup of that query block) of the split scratch.  Every workgroup then does the
// same number of tile steps whatever m, n and the number of compute units are -- a reservation that leaves
// room for RCCL's kernels, an odd m or a short shard no longer break "the grid is exactly one round".
// Slabs a query block does not use are filled with the empty triple (0, -inf, 0) by its last piece, so
// that the merge passes (split_merge_kernel, the hosts' slot merge) stay what they are.  When the cuts
// coincide with the classic equal splits the pieces, their slabs and therefore the results are the same
// bit for bit.
#define SDPA_PK_SK 0
#define SDPA_PK_STREAM 0
#include "sdpa_fwd_f32_pipelined.inc"
#undef SDPA_PK_SK
#define SDPA_PK_SK 1
#include "sdpa_fwd_f32_pipelined.inc"
#undef SDPA_PK_SK
#undef SDPA_PK_STREAM
#define SDPA_PK_SK 0
#define SDPA_PK_STREAM 1
#include "sdpa_fwd_f32_pipelined.inc"
#undef SDPA_PK_SK
#undef SDPA_PK_STREAM

// ---------------------------------------------------------------------------
// In-GPU split merge: the reference's shard merge (attention-mpi.c:340-362 minus
// the final 1/gsum, which stays with the caller) applied to the kv_splits partial
// triples of one GPU.  One thread per (row, 4 columns).
// ---------------------------------------------------------------------------
__global__ void split_merge_kernel(PartialArgs a) {
    const int c4n = (a.dv + 3) / 4;            // real columns only: contrib's rows may be narrower than the slots'
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.m * c4n) return;
    const int row = (int)(idx / c4n), c4 = (int)(idx % c4n);
    float gm = -INFINITY;
    for (int s = 0; s < a.kv_splits; ++s) gm = fmaxf(gm, a.ws_lmax[(size_t)s * a.ws_rows + row]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float tot = 0.f;
    for (int s = 0; s < a.kv_splits; ++s) {
        const float lm = a.ws_lmax[(size_t)s * a.ws_rows + row];
        const float w = (lm == -INFINITY) ? 0.f : expf(lm - gm);
        tot = fmaf(w, a.ws_lsum[(size_t)s * a.ws_rows + row], tot);
        {
            const float4 o = *reinterpret_cast<const float4 *>(
                a.ws_contrib + ((size_t)s * a.ws_rows + row) * a.ws_ld + 4 * c4);
            acc.x = fmaf(w, o.x, acc.x); acc.y = fmaf(w, o.y, acc.y);
            acc.z = fmaf(w, o.z, acc.z); acc.w = fmaf(w, o.w, acc.w);
        }
    }
    *reinterpret_cast<float4 *>(a.contrib + (size_t)row * a.ldo + 4 * c4) = acc;
    if (c4 == 0) {
        a.lmax[row] = gm;
        a.lsum[row] = tot;
    }
}

// The same merge with the single-shard FINISH fused in (round 6): merge step 5 with gsum = the merged lsum (attention-mpi.c:358-362)
// and the writeback (:373) -- the rows leave normalised and dense, as fp64 (out64) or as the fp32 the host widens (out32).  One pass
// over the slabs instead of three kernels (split_merge, normalise / finish, f2d) and two 4-MB round trips through contrib.
// Same sums in the same order and the same fp32 product x * (1 / tot) as finish_f64_kernel / finish_f32_kernel: the rows are the
// separate kernels' bit for bit.
__global__ void split_merge_finish_kernel(PartialArgs a, FinishTarget f) {
    const int c4n = (a.dv + 3) / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.m * c4n) return;
    const int row = (int)(idx / c4n), c4 = (int)(idx % c4n);
    float gm = -INFINITY;
    for (int s = 0; s < a.kv_splits; ++s) gm = fmaxf(gm, a.ws_lmax[(size_t)s * a.ws_rows + row]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float tot = 0.f;
    for (int s = 0; s < a.kv_splits; ++s) {
        const float lm = a.ws_lmax[(size_t)s * a.ws_rows + row];
        const float w = (lm == -INFINITY) ? 0.f : expf(lm - gm);
        tot = fmaf(w, a.ws_lsum[(size_t)s * a.ws_rows + row], tot);
        const float4 o = *reinterpret_cast<const float4 *>(a.ws_contrib + ((size_t)s * a.ws_rows + row) * a.ws_ld + 4 * c4);
        acc.x = fmaf(w, o.x, acc.x); acc.y = fmaf(w, o.y, acc.y);
        acc.z = fmaf(w, o.z, acc.z); acc.w = fmaf(w, o.w, acc.w);
    }
    const float inv = (tot == 0.f) ? 0.f : 1.0f / tot;
    const float v[4] = {acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv};
    const size_t at = (size_t)row * a.dv + 4 * c4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (4 * c4 + i < a.dv) {
            if (f.out64) f.out64[at + i] = (double)v[i];
            if (f.out32) f.out32[at + i] = v[i];
        }
    if (c4 == 0 && a.lmax) {          // (the statistics, for callers that look at them)
        a.lmax[row] = gm;
        a.lsum[row] = tot;
    }
}

// ---------------------------------------------------------------------------
// Any-shape kernel (dk or dv > 128): one wave per query row, lanes across key
// rows for the dot products and across value columns for the accumulate.  A
// correctness path for shapes the MFMA kernel does not cover in fp32; slow.
// ---------------------------------------------------------------------------
constexpr int kGenericMaxCols = 16;   // dv <= 64 * 16
static const int kGenericMaxColsAnchor = 0;   // (an address of this library's own image: see next_ticket_tag)

// q_in_lds = 0 (round 5: dk beyond what four Q rows of LDS hold, > 4096): the lanes read the query row from global
// memory instead -- every lane the same address, served by the L1 -- so that NO dk is refused: the reference's
// dot_avx512 loops over any n (attention-mpi.c:103-121), a drop-in must answer too, at whatever rate.
__global__ __launch_bounds__(256) void generic_partial_kernel(PartialArgs a, float scale, int q_in_lds) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    const float *qs = a.Q + (size_t)(row < a.m ? row : 0) * a.ldq;
    if (q_in_lds) {
        float *ql = smem + (size_t)wave * a.ldq;
        if (row < a.m)
            for (int t = lane; t < a.ldq; t += 64) ql[t] = a.Q[(size_t)row * a.ldq + t];
        qs = ql;
    }
    __syncthreads();
    if (row >= a.m) return;

    float acc[kGenericMaxCols];
#pragma unroll
    for (int cidx = 0; cidx < kGenericMaxCols; ++cidx) acc[cidx] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int dk4 = (a.dk + 3) / 4;   // pad columns are zero on both operands

    for (int kv0 = 0; kv0 < a.n_local; kv0 += 64) {
        const int j = kv0 + lane;
        float s = -INFINITY;
        if (j < a.n_local) {
            const float4 *kp = reinterpret_cast<const float4 *>(a.K + (size_t)j * a.ldk);
            const float4 *qp = reinterpret_cast<const float4 *>(qs);
            float d = 0.f;
            for (int t = 0; t < dk4; ++t) {
                const float4 kk = kp[t], qq = qp[t];
                d = fmaf(kk.x, qq.x, d); d = fmaf(kk.y, qq.y, d);
                d = fmaf(kk.z, qq.z, d); d = fmaf(kk.w, qq.w, d);
            }
            s = d * scale;
        }
        float tmax = s;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - m_new);
        const float p = (j < a.n_local) ? expf(s - m_new) : 0.f;
        float psum = p;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) psum += __shfl_xor(psum, o);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int cidx = 0; cidx < kGenericMaxCols; ++cidx) acc[cidx] *= alpha;
        const int cnt = min(64, a.n_local - kv0);
        for (int jj = 0; jj < cnt; ++jj) {
            const float pj = __shfl(p, jj);
            const float *vrow = a.V + (size_t)(kv0 + jj) * a.ldv;
#pragma unroll
            for (int cidx = 0; cidx < kGenericMaxCols; ++cidx) {
                const int col = lane + 64 * cidx;
                if (col < a.dv) acc[cidx] = fmaf(pj, vrow[col], acc[cidx]);
            }
        }
    }
#pragma unroll
    for (int cidx = 0; cidx < kGenericMaxCols; ++cidx) {
        const int col = lane + 64 * cidx;
        if (col < a.dv) a.contrib[(size_t)row * a.ldo + col] = acc[cidx];
    }
    if (lane == 0) {
        a.lmax[row] = m_run;
        a.lsum[row] = l_run;
    }
}

// ---------------------------------------------------------------------------
// host-side launch logic
// ---------------------------------------------------------------------------
static inline int pad_dim(int d) { return d <= 32 ? 32 : (d <= 64 ? 64 : (d <= 128 ? 128 : 256)); }
static inline int dv_chunk(int dv) { return dv <= 32 ? 32 : (dv <= 64 ? 64 : 128); }
static inline int dv_chunks(int dv) { return (dv + dv_chunk(dv) - 1) / dv_chunk(dv); }


// the dk-split kernel takes 256 < dk <= 512, and 128 < dk <= 256 when dv needs more than one
// 128-column chunk of fused_partial_kernel (measured: 102 vs 86 TFLOP/s at dk = dv = 256, but
// 74 vs 118 at dk = 256, dv = 64, where its per-tile exchange is not amortised)
static inline bool uses_dksplit(int dk, int dv) {
    return dk <= kMaxDkSplit && (dk > kMaxMfmaDk || (dk > kMaxFastDim && dv > kMaxFastDim));
}

// dense head dims whose operand images run fused_pipelined_kernel (launch_shard_partial below): both in
// (32, 128], or one in (128, 256] with the other in (64, 256]
static inline bool pipelined_dims(int dk, int dv) {
    if (dk <= 32 || dv <= 32 || dk > 256 || dv > 256) return false;
    const int kp = dense_ld(dk), vp = dense_ld(dv);
    return !(kp == 256 && vp == 64) && !(kp == 64 && vp == 256);
}
// ... and of those the ones with a stream-K instantiation: dk <= 128.  The 256-wide-dk kernels hold a 128-register
// Q fragment beside the full accumulator file; the piece loop's few extra live values make hipcc spill inside
// their steady-state loop (tests/test_kernel_isa.py), so they keep the classic grid.
static inline bool streamk_dims(int dk, int dv) { return pipelined_dims(dk, dv) && dk <= kMaxFastDim; }

F32Plan plan_f32_launch(int m, int n_local, int dk, int dv, int cus) {
    F32Plan p = {1, 0, 0, 0};
    if (dk > kMaxDkSplit) return p;              // VALU-only fallback kernel: no splits
    if (m <= 0 || n_local <= 0) return p;
    if (cus <= 0) cus = kChipCus;
    const int ntiles = (n_local + kKvTile - 1) / kKvTile;
    int cap = ntiles / 4;                        // >= 4 tiles a split
    if (cap < 1) cap = 1;
    // head dims in (128, 256] x (.., 256]: the dense images (what the hosts always build) take the pipelined
    // kernels at one 128-row workgroup per CU, whatever uses_dksplit() says about other leading dimensions
    const bool wide256 = dk <= kMaxMfmaDk && dv <= 256 && (dk > kMaxFastDim || dv > kMaxFastDim);
    long blocks;
    int slots;
    double rate;
    if (uses_dksplit(dk, dv) && !wide256) {      // 64-row workgroups (32 at dk > 512), one per CU
        blocks = (long)((m + dksplit_rows(dk) - 1) / dksplit_rows(dk)) * dksplit_chunks(dk, dv);
        slots = cus;
        rate = 1.0e14;
    } else {
        blocks = (long)((m + kQRowsPerBlock - 1) / kQRowsPerBlock) * (wide256 ? 1 : dv_chunks(dv));
        slots = (dk > kMaxFastDim || wide256) ? cus : 2 * cus;   // 2 resident workgroups per CU, 1 beyond 128-wide operands
        rate = slots == 2 * cus ? 1.4e14 : 1.3e14;
    }
    rate *= (double)cus / kChipCus;
    int want = (int)((slots + blocks - 1) / blocks);
    if (want > cap) want = cap;
    if (want > 64) want = 64;
    if (want < 1) want = 1;
    // more than one round of workgroups: a fuller last round (sdpa_internal.h)
    const double kernel_s = 2.0 * m * (double)n_local * (dk + dv) / rate;
    const double slab_s = 2.0 * m * (double)dense_ld(dv) * sizeof(float) / 3.0e12;
    p.splits = splits_for_full_rounds(blocks, slots, want, cap, kernel_s, slab_s);

    // ---- stream-K instead?  (pipelined kernels only; $SDPA_DEBUG streamk = 0 / 1 / auto)
    const int knob = launch_knobs().streamk;
    if (knob == 0 || !streamk_dims(dk, dv)) return p;
    const long total = blocks * ntiles;          // tile steps of the launch (blocks = query blocks here)
    if (total >= (1L << 31) - 4096 || blocks > 65536) return p;
    int workers = (int)std::min<long>(slots, std::max<long>(1, total / 4));
    int run = (int)((total + workers - 1) / workers);
    run = std::max(run, (ntiles + 61) / 62);     // at most 64 pieces (slabs) per query block, as the classic splits
    workers = (int)((total + run - 1) / run);
    int pieces = 1;                              // the most pieces a query block is cut into
    for (long q = 0; q < blocks; ++q) {
        const int first = (int)(q * ntiles / run), last = (int)(((q + 1) * ntiles - 1) / run);
        pieces = std::max(pieces, last - first + 1);
    }
    // cost in tile steps per workgroup slot (+3: a piece's prologue and epilogue, a worker has at most
    // run / ntiles + 2 pieces), then the slabs the merge reads back
    const double step_s = kernel_s * slots / (double)total;
    const long wg = blocks * p.splits, rounds = (wg + slots - 1) / slots;
    const double t_classic = rounds * ((ntiles + p.splits - 1) / p.splits) * step_s + (p.splits > 1 ? p.splits * slab_s : 0.0);
    const double t_sk = (run + 3.0 * (run / ntiles + 1)) * step_s + (pieces > 1 ? pieces * slab_s : 0.0);
    if (knob == 1 || t_sk < t_classic) {
        p.splits = pieces;
        p.streamk = 1;
        p.workers = workers;
        p.run = run;
    }
    return p;
}

int pick_kv_splits(int m, int n_local, int dk, int dv, int cus) { return plan_f32_launch(m, n_local, dk, dv, cus).splits; }

size_t workspace_bytes_for(int m, int dv, int splits) {
    if (splits <= 1) return 0;
    const size_t ws_ld = (size_t)dense_ld(dv);         // the padded kernels write whole 64/128/256-column rows
    const size_t tickets = (size_t)((m + kQRowsPerBlock - 1) / kQRowsPerBlock) * sizeof(unsigned long long);
    return (size_t)splits * (size_t)m * (ws_ld + 2) * sizeof(float) + tickets;
}

// Scratch for the launch on ANY stream: the most slabs the plan asks for over the whole chip and every reservation
// create_masked_stream() accepts (8, 16, ... CUs left out, up to half the chip).
size_t workspace_bytes(int m, int n_local, int dk, int dv) {
    // (17 launch plans per call, and the C host's planner asks ~40 times per problem: 0.2 ms of every sdpa_attention_f64 call at the
    //  metric shape went here before its first launch was enqueued -- round 6, profiles/r06/config2_boundary_steps.log -- so the answers
    //  are remembered; the only knob the plans read is the stream-K one)
    struct Key { int m, n, dk, dv, knob; size_t bytes; };
    static std::mutex mu;
    static std::vector<Key> seen;
    const int knob = launch_knobs().streamk;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Key &k : seen)
            if (k.m == m && k.n == n_local && k.dk == dk && k.dv == dv && k.knob == knob) return k.bytes;
    }
    int s = 1;
    for (int cus = kChipCus; cus >= kChipCus / 2; cus -= 8) s = std::max(s, pick_kv_splits(m, n_local, dk, dv, cus));
    const size_t bytes = workspace_bytes_for(m, dv, s);
    std::lock_guard<std::mutex> lk(mu);
    if (seen.size() >= 256) seen.clear();
    seen.push_back({m, n_local, dk, dv, knob, bytes});
    return bytes;
}

void carve_workspace(PartialArgs &a, void *ws, int ws_ld) {
    a.ws_ld = ws_ld;
    a.ws_contrib = (float *)ws;
    a.ws_lmax = a.ws_contrib + (size_t)a.kv_splits * a.m * ws_ld;
    a.ws_lsum = a.ws_lmax + (size_t)a.kv_splits * a.m;
    // (ws_ld + 2) is even, so the arrival words are 8-byte aligned whenever the area is
    a.tickets = reinterpret_cast<unsigned long long *>(a.ws_lsum + (size_t)a.kv_splits * a.m);
}

// Generation of a launch's arrival words (PartialArgs::ticket_tag): 56 bits, never zero, never repeated
// within a process, and started from a per-process value so that two copies of this library in one
// process (tools/ A/B runs) do not hand the same numbers to launches that share a scratch area.
static unsigned long long next_ticket_tag() {
    static std::atomic<unsigned long long> gen{
        ((unsigned long long)time(nullptr) << 24) ^ ((unsigned long long)(uintptr_t)&kGenericMaxColsAnchor << 4)};
    unsigned long long t;
    do t = gen.fetch_add(1, std::memory_order_relaxed) & ((1ull << 56) - 1); while (t == 0);
    return t;
}

hipError_t launch_split_merge(const PartialArgs &a_in, hipStream_t s) {
    PartialArgs a = a_in;
    if (a.ws_rows <= 0) a.ws_rows = a.m;
    const long work = (long)a.m * ((a.dv + 3) / 4);
    hipLaunchKernelGGL(split_merge_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_split_merge_finish(const PartialArgs &a_in, const FinishTarget &f, hipStream_t s) {
    PartialArgs a = a_in;
    if (a.ws_rows <= 0) a.ws_rows = a.m;
    if (!f.out64 && !f.out32) return hipErrorInvalidValue;
    const long work = (long)a.m * ((a.dv + 3) / 4);
    hipLaunchKernelGGL(split_merge_finish_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, a, f);
    return hipGetLastError();
}

template <int DKP, int DVP>
static hipError_t launch_fast(const PartialArgs &a, hipStream_t s) {
    const int nqb = (a.m + kQRowsPerBlock - 1) / kQRowsPerBlock;
    const int ntiles = (a.n_local + kKvTile - 1) / kKvTile;
    const int tiles_per_split = (ntiles + a.kv_splits - 1) / a.kv_splits;
    const int kv_per_split = tiles_per_split > 0 ? tiles_per_split * kKvTile : kKvTile;
    const int chunks = dv_chunks(a.dv);
    const size_t lds = (size_t)2 * kKvTile * ((DKP + 4) + DVP) * sizeof(float);
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&fused_partial_kernel<DKP, DVP>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[dev].store(true, std::memory_order_release);
    }
    const float scale = 1.0f / sqrtf((float)a.dk);   // attention-mpi.c:208
    hipLaunchKernelGGL((fused_partial_kernel<DKP, DVP>), dim3(nqb * chunks * a.kv_splits), dim3(256), lds,
                       s, a, kv_per_split, nqb, chunks, scale);
    note_launch("fused_partial_kernel", 2, DKP, DVP, 0, 0, 0, nqb * chunks * a.kv_splits, a.kv_splits, 0, a.m, a.n_local);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.kv_splits > 1 && !a.defer_merge) e = launch_split_merge(a, s);
    return e;
}

// per-device "dynamic LDS size raised" flags of one kernel instantiation (set from any enqueue thread)
struct AttrOnce {
    std::atomic<bool> done[64];
    AttrOnce() { for (auto &d : done) d.store(false); }
    template <typename K> hipError_t ensure(K kernel, int dev, size_t lds) {
        if (done[dev].load(std::memory_order_acquire)) return hipSuccess;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) done[dev].store(true, std::memory_order_release);
        return e;
    }
};

template <int DK, int DV, int ABL = 0>
static hipError_t launch_pipelined(const PartialArgs &a, hipStream_t s) {
    const int nqb = (a.m + kQRowsPerBlock - 1) / kQRowsPerBlock;
    const int ntiles = (a.n_local + kKvTile - 1) / kKvTile;
    const int tiles_per_split = (ntiles + a.kv_splits - 1) / a.kv_splits;
    const int kv_per_split = tiles_per_split > 0 ? tiles_per_split * kKvTile : kKvTile;
    const size_t lds = (size_t)2 * kKvTile * (DK + DV) * sizeof(float);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    const float scale = 1.0f / sqrtf((float)a.dk);   // attention-mpi.c:208
    // kv_splits > 1: the partial triples are merged by split_merge_kernel right behind.  The kernel can
    // also merge them itself ($SDPA_DEBUG split_merge=kernel: the last workgroup of a query
    // block to arrive does it, one launch per step) -- same sums in the same order, bit for bit
    // (tests/test_gpu_parity.py) -- but measured SLOWER on MI355X (profiles/r03/split_merge_forms_ab.log:
    // config 2 0.301 vs 0.273 ms per step, a 1/8 rank share 1.013 vs 0.993): the separate pass spreads
    // the 34 MB of slab reads over every CU, the last arrivers are 64 workgroups in a latency-bound tail.
    PartialArgs k = a;
    if (k.kv_splits <= 1 || k.defer_merge || (reinterpret_cast<uintptr_t>(k.tickets) & 7) != 0 ||
        !launch_knobs().split_merge_kernel)
        k.tickets = nullptr;
    hipError_t e;
    // stream-K (sdpa_internal.h: F32Plan): the plan for THIS stream's compute units; taken when the caller gave
    // the launch at least the slabs it needs (callers size a.kv_splits with the same function)
    const F32Plan plan = ABL || k.tickets ? F32Plan{1, 0, 0, 0} : plan_f32_launch(a.m, a.n_local, a.dk, a.dv, a.cus > 0 ? a.cus : stream_cus(s));
    bool launched = false;
    if constexpr (DK <= kMaxFastDim) {
        if (plan.streamk && plan.splits <= a.kv_splits && (a.kv_splits <= 1 || a.ws_contrib)) {
            static AttrOnce attr;
            if ((e = attr.ensure(&fused_pipelined_sk_kernel<DK, DV>, dev, lds)) != hipSuccess) return e;
            hipLaunchKernelGGL((fused_pipelined_sk_kernel<DK, DV>), dim3(plan.workers), dim3(256), lds, s, k,
                               plan.run, nqb, scale);
            note_launch("fused_pipelined_sk_kernel", 2, DK, DV, 0, 0, 0, plan.workers, plan.splits, 1, a.m, a.n_local);
            launched = true;
        }
    }
    if (launched) {
    } else if (k.tickets) {
        k.ticket_tag = next_ticket_tag();
        static AttrOnce attr;
        if ((e = attr.ensure(&fused_pipelined_kernel<DK, DV, ABL, 1>, dev, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((fused_pipelined_kernel<DK, DV, ABL, 1>), dim3(nqb * k.kv_splits), dim3(256), lds, s,
                           k, kv_per_split, nqb, scale);
        note_launch("fused_pipelined_kernel", 4, DK, DV, ABL, 1, 0, nqb * k.kv_splits, k.kv_splits, 0, a.m, a.n_local);
    } else {
        static AttrOnce attr;
        if ((e = attr.ensure(&fused_pipelined_kernel<DK, DV, ABL, 0>, dev, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((fused_pipelined_kernel<DK, DV, ABL, 0>), dim3(nqb * k.kv_splits), dim3(256), lds, s,
                           k, kv_per_split, nqb, scale);
        note_launch("fused_pipelined_kernel", 4, DK, DV, ABL, 0, 0, nqb * k.kv_splits, k.kv_splits, 0, a.m, a.n_local);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (k.kv_splits > 1 && !k.defer_merge && !k.tickets) e = launch_split_merge(k, s);
    return e;
}

bool stream_launch_supported(int dk, int dv) { return dk > 32 && dv > 32 && dk <= kMaxFastDim && dv <= kMaxFastDim; }

template <int DK, int DV>
static hipError_t launch_streamed(const PartialArgs &a, const StreamArgs &st, hipStream_t s) {
    const int nqb = (a.m + kQRowsPerBlock - 1) / kQRowsPerBlock;
    const int ntiles = (a.n_local + kKvTile - 1) / kKvTile;
    const int tiles_per_split = (ntiles + a.kv_splits - 1) / a.kv_splits;
    const int kv_per_split = tiles_per_split > 0 ? tiles_per_split * kKvTile : kKvTile;
    const size_t lds = (size_t)2 * kKvTile * (DK + DV) * sizeof(float);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    const float scale = 1.0f / sqrtf((float)a.dk);   // attention-mpi.c:208
    PartialArgs k = a;
    k.tickets = nullptr;
    static AttrOnce attr;
    hipError_t e;
    if ((e = attr.ensure(&fused_pipelined_stream_kernel<DK, DV>, dev, lds)) != hipSuccess) return e;
    hipLaunchKernelGGL((fused_pipelined_stream_kernel<DK, DV>), dim3(nqb * k.kv_splits), dim3(256), lds, s, k, kv_per_split,
                       nqb, scale, st);
    note_launch("fused_pipelined_stream_kernel", 2, DK, DV, 0, 0, 0, nqb * k.kv_splits, k.kv_splits, 0, a.m, a.n_local);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (k.kv_splits > 1 && !k.defer_merge) e = launch_split_merge(k, s);
    return e;
}

hipError_t launch_shard_partial_streamed(const PartialArgs &a_in, const StreamArgs &st, hipStream_t s) {
    PartialArgs a = a_in;
    if (a.ws_rows <= 0) a.ws_rows = a.m;
    a.tune = 0;
    const int kp = a.ldq, vp = a.ldv;
    const bool dense = a.ldq == a.ldk && (kp == 64 || kp == 128) && kp >= a.dk && (vp == 64 || vp == 128) && vp >= a.dv &&
                       a.ldo >= vp && a.ldo % 4 == 0 && (a.kv_splits <= 1 || (a.ws_contrib && a.ws_ld >= vp)) &&
                       (reinterpret_cast<uintptr_t>(a.K) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.V) & 15) == 0;
    if (!dense || !st.flags || !st.status || !st.abort || st.n_chunks < 1 || st.n_chunks > kStreamMaxChunks || a.n_local <= 0)
        return hipErrorInvalidValue;
    if (kp == 128 && vp == 128) return launch_streamed<128, 128>(a, st, s);
    if (kp == 64 && vp == 64) return launch_streamed<64, 64>(a, st, s);
    if (kp == 128 && vp == 64) return launch_streamed<128, 64>(a, st, s);
    return launch_streamed<64, 128>(a, st, s);
}

hipError_t launch_shard_partial(const PartialArgs &a_in, hipStream_t s) {
    PartialArgs a = a_in;
    if (a.ws_rows <= 0) a.ws_rows = a.m;
#ifdef SDPA_ABLATIONS   // tools/ builds only: the shipped library never reads $SDPA_DEBUG tune
    static const int tune_env = sdpa_debug_int("tune", 0);
    a.tune = tune_env;
#else
    a.tune = 0;
#endif
    // Operand images whose rows are 64 / 128 / 256 floats wide (dims padded with zero columns, the
    // header's contract for columns [dk, ld)) take the software-pipelined LDS-DMA kernel of that width:
    // same padded MFMA work as the any-shape kernels, at the pipelined kernel's rate.  The softmax
    // scale is 1/sqrt of the TRUE dk; contrib rows must hold the padded width ($SDPA_DEBUG tune&4: off).
    const int kp = a.ldq, vp = a.ldv;
    const bool dense = a.ldq == a.ldk && (kp == 64 || kp == 128 || kp == 256) && kp >= a.dk &&
                       (vp == 64 || vp == 128 || vp == 256) && vp >= a.dv && a.ldo >= vp && a.ldo % 4 == 0 &&
                       (a.kv_splits <= 1 || a.ws_ld >= vp) &&
                       (reinterpret_cast<uintptr_t>(a.K) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.V) & 15) == 0;
    if (dense && !(a.tune & 4)) {       // one wave per SIMD: Q (128 VGPRs at dk = 256) and O^T (128 AGPRs at dv = 256) resident
        if (kp == 256 && vp == 256) return launch_pipelined<256, 256>(a, s);
        if (kp == 256 && vp == 128) return launch_pipelined<256, 128>(a, s);
        if (kp == 128 && vp == 256) return launch_pipelined<128, 256>(a, s);
    }
    if (uses_dksplit(a.dk, a.dv) && !(a.tune & 8))     // $SDPA_DEBUG tune&8: the kernels it replaced
        return launch_dksplit(a, s);                    // sdpa_fwd_f32_dksplit.hip
    if (a.dk > kMaxMfmaDk) {
        size_t lds = (size_t)4 * a.ldq * sizeof(float);
        const int q_in_lds = lds <= 64 * 1024 ? 1 : 0;             // dk <= 4096: the workgroup's four Q rows live in LDS
        if (!q_in_lds) lds = 16;
        const float scale = 1.0f / sqrtf((float)a.dk);
        // the kernel holds 64 * kGenericMaxCols value columns per row: wider V goes in column chunks, one
        // launch each (the scores are recomputed; lmax / lsum come out the same from every chunk)
        for (int c0 = 0; c0 < a.dv; c0 += 64 * kGenericMaxCols) {
            PartialArgs ac = a;
            ac.V = a.V + c0;
            ac.contrib = a.contrib + c0;
            ac.dv = std::min(64 * kGenericMaxCols, a.dv - c0);
            hipLaunchKernelGGL(generic_partial_kernel, dim3((a.m + 3) / 4), dim3(256), lds, s, ac, scale, q_in_lds);
            note_launch("generic_partial_kernel", 0, 0, 0, 0, 0, 0, (a.m + 3) / 4, 1, 0, a.m, a.n_local);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    if (dense && !(a.tune & 4)) {
        if (kp == 128 && vp == 128) {
#ifdef SDPA_ABLATIONS
            switch ((a.tune >> 4) & 7) {     // timing-only ablations, see fused_pipelined_kernel
                case 1: return launch_pipelined<128, 128, 1>(a, s);
                case 2: return launch_pipelined<128, 128, 2>(a, s);
                case 3: return launch_pipelined<128, 128, 3>(a, s);
                case 4: return launch_pipelined<128, 128, 4>(a, s);
                case 7: return launch_pipelined<128, 128, 7>(a, s);
                default: break;
            }
#endif
            return launch_pipelined<128, 128>(a, s);
        }
        if (kp == 64 && vp == 64) return launch_pipelined<64, 64>(a, s);
        if (kp == 128 && vp == 64) return launch_pipelined<128, 64>(a, s);
        if (kp == 64 && vp == 128) return launch_pipelined<64, 128>(a, s);
    }
    const int kpad = pad_dim(a.dk), vchunk = dv_chunk(a.dv);
#define SDPA_CASE(KP, VP) if (kpad == KP && vchunk == VP) return launch_fast<KP, VP>(a, s);
    SDPA_CASE(256, 128) SDPA_CASE(256, 64) SDPA_CASE(256, 32)
    SDPA_CASE(128, 128) SDPA_CASE(128, 64) SDPA_CASE(128, 32)
    SDPA_CASE(64, 128)  SDPA_CASE(64, 64)  SDPA_CASE(64, 32)
    SDPA_CASE(32, 128)  SDPA_CASE(32, 64)  SDPA_CASE(32, 32)
#undef SDPA_CASE
    return hipErrorInvalidValue;
}

// (sdpa_internal.h: preload_kernels_*) touching one kernel makes the runtime load this translation unit's code object for the
// current device NOW -- not in front of the first launch that needs it, possibly behind a resident persistent launch
hipError_t preload_kernels_f32() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&split_merge_kernel));
}

}  // namespace sdpa
