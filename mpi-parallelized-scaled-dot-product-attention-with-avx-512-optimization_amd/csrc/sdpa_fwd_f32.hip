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
//   (fused_dksplit_kernel<DKS,DVS,QB>, 256 < dk <= 1024 and non-dense 128 < dk <= 256 with dv > 128, lives in
//    sdpa_fwd_f32_dksplit.hip)
//   generic_partial_kernel          dk > 1024: VALU-only correctness path
//   split_merge_kernel              merge of the in-GPU K/V splits
//
#include "sdpa_f32_device.h"

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
// ABL: timing-only ablation switches (results are wrong when non-zero; $SDPA_TUNE selects them):
//   1 = no DMA / no barrier in the steady state, 2 = no LDS fragment reads, 4 = no softmax VALU
// MERGE: 1 = the launch merges its K/V splits itself (arrival words, $SDPA_SPLIT_MERGE=kernel); the shipped
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
template <int DK, int DV, int ABL = 0, int MERGE = 0, int SK = 0>
__global__ __launch_bounds__(256, (DK + DV > 256) ? 1 : 2) void fused_pipelined_kernel(PartialArgs a, int kv_per_split,
                                                                                       int n_qblocks, float scale) {
    static_assert(!(SK && MERGE), "the in-kernel split merge counts equal splits");
    constexpr int NU = DK / 8;                // 16-byte K reads per tile per lane
    constexpr int PPU = NU <= 16 ? 16 / NU : 1;   // P values finished per 4-MFMA QK^T step ...
    constexpr int PEV = NU <= 16 ? 1 : NU / 16;   // ... of every PEV-th step (dk = 256: every other one)
    constexpr int NT = DV / 32;               // O^T tiles
    // V columns of a lane: NV consecutive floats per vector read, NH reads 128 columns apart.  O^T tile
    // tt, row i is V column (tt / NV) * 32 NV + NV i + tt % NV  (dv <= 128: NT i + tt)
    constexpr int NV = NT < 4 ? NT : 4;
    constexpr int NH = NT / NV;
    // One wave per SIMD with the whole 512-register file (dk + dv > 256): O^T lives in the accumulator
    // file and is touched by nothing but MFMAs and "+a" asm (any plain VALU use of it makes hipcc shuttle
    // tiles between the files on the hot path, as in the bf16 wide kernel); the score chains are inline-asm
    // MFMAs with VGPR C/D, because in this mode hipcc puts every builtin MFMA result in AGPRs and the
    // softmax works on VGPRs.  hipcc sees no MFMA inside an asm statement: wait states are placed by hand.
    constexpr bool WIDE = DK + DV > 256;
    constexpr int KTILE = kKvTile * DK;       // floats
    constexpr int VTILE = kKvTile * DV;
    constexpr int KCH = DK / 4;               // 16-byte chunks per K row
    constexpr int VCH = DV / 4;
    constexpr int KPW = (kKvTile * KCH / 64) / 4;   // 1-KiB DMA pieces per wave per K tile
    constexpr int VPW = (kKvTile * VCH / 64) / 4;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *const Ks = smem;                   // [2][KTILE], swizzled chunks
    float *const Vs = smem + 2 * KTILE;       // [2][VTILE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int hi = lane >> 5;

    const int work = xcd_remap(blockIdx.x, gridDim.x);
    // the piece of work in hand: query block, K/V range, slab.  Classic: fixed for the launch.  SK: one per
    // trip of the piece loop below (sk_pos .. sk_end = this workgroup's run of tile steps)
    int split, qblock, qrow, kv_begin, kv_end, T;
    int sk_pos = 0, sk_end = 0, sk_ntiles = 1;
    if constexpr (SK) {
        sk_ntiles = (a.n_local + kKvTile - 1) / kKvTile;
        const int total = n_qblocks * sk_ntiles;              // < 2^31: the launcher checks
        sk_pos = min(total, work * kv_per_split);             // kv_per_split = tile steps per workgroup here
        sk_end = min(total, sk_pos + kv_per_split);
        if (sk_pos >= sk_end) return;
        split = 0; qblock = 0; qrow = 0; kv_begin = 0; kv_end = 0; T = 0;
    } else {
        split = work / n_qblocks;
        qblock = work - split * n_qblocks;
        qrow = qblock * kQRowsPerBlock + wave * 32 + li;
        kv_begin = split * kv_per_split;
        kv_end = min(a.n_local, kv_begin + kv_per_split);
        T = kv_end > kv_begin ? (kv_end - kv_begin + kKvTile - 1) / kKvTile : 0;
    }
    const float c = scale * 1.44269504088896340736f;

    // Q is pre-multiplied by scale*log2(e): the MFMA chain then yields scores directly in the
    // exp2 domain, and with the accumulator initialised to -m_ref it yields (score - m_ref), so a
    // P value costs ONE VALU instruction (v_exp_f32).  VALU cycles are MFMA cycles lost here: the
    // f32-input MFMA runs at the f32 vector rate and does not overlap VALU issue.
    float4 qf[NU];
    auto load_q = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            qf[u] = qrow < a.m ? *reinterpret_cast<const float4 *>(a.Q + (size_t)qrow * DK + 8 * u + 4 * hi)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
            qf[u].x *= c; qf[u].y *= c; qf[u].z *= c; qf[u].w *= c;
        }
    };
    if constexpr (!SK) load_q();

    f32x16 oacc[NT];
    // Softmax state of this lane's query row, all in the exp2 domain:
    //   m_ref   reference exponent the accumulators are relative to (the row max when it was
    //           last moved; NOT moved for rises below `defer` -- fp32 has the headroom)
    //   max_rel running (true row max - m_ref) >= 0, folded back in at the epilogue
    //   l_run   this half-wave's share of sum exp2(score - m_ref)
    // Deferring spends up to 2^kDeferLog2 of fp32's exponent range: weights reach 2^24 instead of 1,
    // so for |V| * n_local beyond ~2^104 the un-normalised sums overflow where the reference's eager
    // rescale (attention-mpi.c:179-182) stays finite.  The epilogue checks: a workgroup that finds a
    // non-finite value in its triple runs its K/V range a second time with defer = 0 (every rise of a
    // row max moves the reference, weights <= 1: the reference's own bound) -- pass 1 below.
    constexpr float kDeferLog2 = 24.0f;
    float defer = kDeferLog2;
    float m_ref, max_rel, l_run;
    auto pin_o = [&]() __attribute__((always_inline)) {
        if constexpr (WIDE) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) asm volatile("" : "+a"(oacc[tt]));
        }
    };
    // one link of a score chain: sx += k * q over two contraction indices
    auto score_link = [&](f32x16 &sx, float kv, float qv) __attribute__((always_inline)) {
        if constexpr (WIDE) {
            // s_nop 1: "VALU write -> MFMA read" needs 2 wait states and the allocator may set an operand up right in front
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(sx) : "v"(kv), "v"(qv));
        } else {
            sx = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qv, sx, 0, 0, 0);
        }
    };
    // a 16-pass MFMA's result may be read by the VALU 18 wait states after issue
    auto score_fence = [&](f32x16 &sx) __attribute__((always_inline)) {
        if constexpr (WIDE) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(sx));
    };
    // ---- LDS-DMA staging: per-lane source byte offsets inside a tile (loop invariant)
    unsigned koff[KPW], voff[VPW];
    {   // (loop invariant, shared by both passes: re-deriving them per pass costs the hot loop two scratch reloads)
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
            const int row = (wave * KPW + j) * (64 / KCH) + lane / KCH;
            const int cpos = lane % KCH;
            koff[j] = (unsigned)(row * DK * 4 + ((cpos ^ (row & 15)) << 4));
        }
#pragma unroll
        for (int j = 0; j < VPW; ++j) {
            const int row = (wave * VPW + j) * (64 / VCH) + lane / VCH;
            voff[j] = (unsigned)(row * DV * 4 + ((lane % VCH) << 4));
        }
    }
    // The DMA is issued from inline asm on purpose: hipcc treats the builtin as a pending LDS
    // write and drains vmcnt(0) before the next ds_read, which would serialise the stream.
    // Hidden in asm, the loads stay in flight under the MFMAs; they are drained by the explicit
    // s_waitcnt vmcnt(0) in front of the end-of-step barrier (stage_fence).
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)smem);
#ifdef SDPA_DMA_ASSERT
    int audit_bad = 0;
#endif
    auto dma_piece = [&](const char *gbase, unsigned lane_off, unsigned lds_byte) __attribute__((always_inline)) {
        if constexpr (ABL & 1) return;
#ifdef SDPA_DMA_ASSERT
        // audit build (tools/build_variant.sh ... -DSDPA_DMA_ASSERT, never shipped): every 16-byte DMA source must lie
        // inside the K or the V image of this launch (the LDS destinations are compile-time offsets of a tile buffer).
        // A violation poisons the row sum (NaN), which every parity test sees -- no branch near the asm.
        {
            const char *src = gbase + lane_off;
            const char *k0 = reinterpret_cast<const char *>(a.K), *k1 = k0 + (size_t)a.n_local * DK * 4;
            const char *v0 = reinterpret_cast<const char *>(a.V), *v1 = v0 + (size_t)a.n_local * DV * 4;
            const bool in_k = src >= k0 && src + 16 <= k1, in_v = src >= v0 && src + 16 <= v1;
            audit_bad |= (!(in_k || in_v) || (lane_off & 15u) != 0) ? 1 : 0;
        }
#endif
        // M0 is written without save/restore: hipcc treats it as reserved and re-initialises it next
        // to each of its own uses (the same choice as in the bf16 wide kernel)
        asm volatile("s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %2"
                     :
                     : "v"(lane_off), "s"(lds_byte), "s"(gbase)
                     : "memory" SDPA_M0_CLOBBER);
    };
    auto stage_fence = [&]() __attribute__((always_inline)) {
        if constexpr (ABL & 1) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // Full tiles (all but possibly the last of a split) take the branch-free path: no per-lane
    // address arithmetic at all in the steady state (VALU cycles are MFMA cycles lost here).
    auto dma_k = [&](int tile, int buf) __attribute__((always_inline)) {
        const int base = kv_begin + tile * kKvTile;
        const int last = kv_end - 1 - base;
        const char *kb = reinterpret_cast<const char *>(a.K + (size_t)base * DK);
        if (last >= kKvTile - 1) {
#pragma unroll
            for (int j = 0; j < KPW; ++j)
                dma_piece(kb, koff[j], lds_base + (unsigned)(buf * KTILE + (wave * KPW + j) * 256) * 4u);
        } else {                               // ragged tile: clamp the source row, keep the chunk
#pragma unroll
            for (int j = 0; j < KPW; ++j) {
                const unsigned row = min((int)(koff[j] / (DK * 4)), last);
                dma_piece(kb, row * (DK * 4) + (koff[j] % (DK * 4)),
                          lds_base + (unsigned)(buf * KTILE + (wave * KPW + j) * 256) * 4u);
            }
        }
    };
    auto dma_v = [&](int tile, int buf) __attribute__((always_inline)) {
        const int base = kv_begin + tile * kKvTile;
        const int last = kv_end - 1 - base;
        const char *vb = reinterpret_cast<const char *>(a.V + (size_t)base * DV);
        if (last >= kKvTile - 1) {
#pragma unroll
            for (int j = 0; j < VPW; ++j)
                dma_piece(vb, voff[j], lds_base + (unsigned)((2 * KTILE + buf * VTILE) + (wave * VPW + j) * 256) * 4u);
        } else {
#pragma unroll
            for (int j = 0; j < VPW; ++j) {
                const unsigned row = min((int)(voff[j] / (DV * 4)), last);
                dma_piece(vb, row * (DV * 4) + (voff[j] % (DV * 4)),
                          lds_base + (unsigned)((2 * KTILE + buf * VTILE) + (wave * VPW + j) * 256) * 4u);
            }
        }
    };

    // K fragment byte addresses inside a K buffer: chunk (2u+hi) of row li, un-swizzled
    // (the XOR only touches the low 4 chunk bits: chunks 16..31 of a 128-wide row are the
    //  same 8 addresses + 256 bytes)
    unsigned kaddr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) kaddr[u] = (unsigned)(li * DK * 4 + (((2 * u + hi) ^ (li & 15)) << 4));

    auto kfrag = [&](int buf, int u) __attribute__((always_inline)) -> float4 {
        if constexpr (ABL & 2) return qf[(u + 1) % NU];
        return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Ks + buf * KTILE) +
                                                 kaddr[u & 7] + (u >> 3) * 256);
    };

    auto mask_ragged = [&](f32x16 &sx, int tile) __attribute__((always_inline)) {
        const int valid = kv_end - (kv_begin + tile * kKvTile);
        if (valid < kKvTile) {
            score_fence(sx);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (crow(r, hi) >= valid) sx[r] = -INFINITY;
        }
    };
    // Row max of a finished score tile (relative to m_ref).  Only a rise of more than
    // 2^defer moves m_ref: O, l and the pending scores `sx` are then all brought to the
    // new reference exactly once.  (Rare: after the first tile it needs a key whose score beats
    // everything seen so far by > 16.6 in natural-log units.)
    auto absorb_rel = [&](float tmax, f32x16 &sx) __attribute__((always_inline)) {
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        if (__any(tmax > defer)) {
            const float jump = fmaxf(tmax, 0.f);
            const float alpha = fast_exp2(-jump);
            if constexpr (WIDE) {           // O stays in the accumulator file: read - scale - write back inside asm
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float tmp;
                        asm volatile("v_accvgpr_read_b32 %1, %0\n\ts_nop 0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1"
                                     : "+a"(oacc[tt][r]), "=&v"(tmp) : "v"(alpha));
                    }
            } else {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[tt][r] *= alpha;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sx[r] -= jump;
            l_run *= alpha;
            m_ref += jump;
            max_rel -= jump;
            tmax -= jump;
        }
        max_rel = fmaxf(max_rel, tmax);
    };

    // one pipelined tile step: consumes S(t) in `su`, produces S(t+1) in `sm`
    auto step = [&](auto has_next, f32x16 &su, f32x16 &sm, int t) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(has_next)::value;
        const int vbuf = t & 1, kbuf = (t + 1) & 1;
        pin_o();
        if (t + 2 < T) dma_k(t + 2, t & 1);
        if (t + 1 < T) dma_v(t + 1, (t + 1) & 1);
        if constexpr (HAS_NEXT) {
            // [A] S^T(t+1) on the matrix pipe  ||  P(t) on the VALU.  Per 4 MFMAs: the K
            // fragment for the step after next is read, and PPU score(s) become P values.  The
            // exp2 is a volatile asm so that it stays inside its sched_barrier-fenced slot.
            float4 kf = kfrag(kbuf, 0);
            float4 kn = kfrag(kbuf, 1);
            {   // accumulator = -m_ref in 8 packed moves instead of 16 (every VALU op beside an
                // f32 MFMA costs matrix-pipe time)
                const f32x2 negm = {-m_ref, -m_ref};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 t2;
                    asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(t2) : "v"(negm));
                    sm[r] = t2.x;
                    sm[r + 1] = t2.y;
                }
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                float4 kn2 = kn;
                if (u + 2 < NU) kn2 = kfrag(kbuf, u + 2);
                __builtin_amdgcn_sched_barrier(0);
                score_link(sm, kf.x, qf[u].x);
                score_link(sm, kf.y, qf[u].y);
                score_link(sm, kf.z, qf[u].z);
                score_link(sm, kf.w, qf[u].w);
#pragma unroll
                for (int r = (u / PEV) * PPU; r < (u % PEV == 0 ? (u / PEV + 1) * PPU : 0); ++r) {
                    if constexpr (ABL & 4) {
                        asm volatile("" : "+v"(su[r]));
                    } else {
                        // exp2 of element r, then the row-sum add of element r-1: the transcendental's
                        // result is never read by the next instruction, so it needs no wait state
                        asm volatile("v_exp_f32 %0, %0" : "+v"(su[r]));
                        if (r > 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l_run) : "v"(su[r - 1]));
                    }
                }
                kf = kn;
                kn = kn2;
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 4)) l_run += su[15];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                su[r] = fast_exp2(su[r]);
                l_run += su[r];
            }
        }

        // [B] O^T += V(t)^T.P(t)^T on the matrix pipe  ||  row max of S^T(t+1) on the VALU
        if constexpr (HAS_NEXT) mask_ragged(sm, t + 1);
        const float *vt = Vs + vbuf * VTILE + NV * li + 4 * hi * DV;
        float tmax = -INFINITY;
        auto vload = [&](int r) __attribute__((always_inline)) -> VFrag<NT> {
            if constexpr (ABL & 2) {
                VFrag<NT> f;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) f.v[tt] = qf[r % NU].x;
                return f;
            } else {
                return VFrag<NT>::load(vt + crow(r, 0) * DV);
            }
        };
        static_assert(NH == 1 || NT == 8, "VFrag<8> reads two float4 128 columns apart");
        VFrag<NT> vf = vload(0);
        VFrag<NT> vn = vload(1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            VFrag<NT> vn2 = vn;
            if (r + 2 < 16) vn2 = vload(r + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
                oacc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.v[tt], su[r], oacc[tt], 0, 0, 0);
            // row max of S(t+1): starts one step late so that the QK^T chain has drained
            if constexpr (HAS_NEXT && !(ABL & 4)) {
                if (r == 2) tmax = pinned_max(sm[0], sm[1]);
                if (r >= 4 && (r & 1) == 0) tmax = pinned_max3(tmax, sm[r - 2], sm[r - 1]);
                if (r == 15) tmax = pinned_max3(tmax, sm[14], sm[15]);
            }
            vf = vn;
            vn = vn2;
        }
        __builtin_amdgcn_sched_barrier(0);
        pin_o();
        if constexpr (HAS_NEXT && !(ABL & 4)) absorb_rel(tmax, sm);
        stage_fence();                        // drain this wave's DMAs, then barrier
    };

    float l_tot = 0.f;
    // one walk over the split's K/V range; instantiated twice (straight-line, no loop around the hot
    // loop: its register allocation stays what it was), the second copy only runs after a failed range check
    auto run_pass = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    pin_o();
    m_ref = 0.f; max_rel = 0.f; l_run = 0.f;
    f32x16 sA, sB;
    if (T > 0) {
        dma_k(0, 0);
        dma_v(0, 0);
        if (T > 1) dma_k(1, 1);
        stage_fence();
        // prologue: S^T(0), its mask and max
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[r] = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const float4 kf = kfrag(0, u);
            score_link(sA, kf.x, qf[u].x);
            score_link(sA, kf.y, qf[u].y);
            score_link(sA, kf.z, qf[u].z);
            score_link(sA, kf.w, qf[u].w);
        }
        score_fence(sA);
        mask_ragged(sA, 0);
        float tmax = sA[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sA[r]);
        m_ref = fmaxf(tmax, __shfl_xor(tmax, 32));      // finite: every tile has a valid key row
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[r] -= m_ref;
        __syncthreads();                      // everyone is done with K(0) before K(2) lands on it

        int t = 0;
        for (; t + 2 < T; t += 2) {
            step(std::true_type(), sA, sB, t);
            step(std::true_type(), sB, sA, t + 1);
        }
        if (T - t == 2) {
            step(std::true_type(), sA, sB, t);
            step(std::false_type(), sB, sA, t + 1);
        } else {
            step(std::false_type(), sA, sB, t);
        }
    }

    // ---- epilogue: express the triple relative to the TRUE row max (m_ref + max_rel), in the
    //      reference's units (lmax is a natural-log score: exp2-domain value * ln 2)
    const float fold = fast_exp2(-max_rel);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[tt][r] *= fold;
    l_tot = (l_run + __shfl_xor(l_run, 32)) * fold;
#ifdef SDPA_DMA_ASSERT
    if (__any(audit_bad)) l_tot = __builtin_nanf("");
#endif
    };  // run_pass

    auto store_rows = [&](float *out, int ldo, float *omax, float *osum, const f32x16 (&o)[NT], float vmax,
                          float vsum) __attribute__((always_inline)) {
        if (qrow < a.m) {
            float *orow = out + (size_t)qrow * ldo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col0 = NV * crow(r, hi);
                if constexpr (NV == 4) {
#pragma unroll
                    for (int h = 0; h < NH; ++h)
                        *reinterpret_cast<float4 *>(orow + 128 * h + col0) =
                            make_float4(o[4 * h][r], o[4 * h + 1][r], o[4 * h + 2][r], o[4 * h + 3][r]);
                } else {
                    *reinterpret_cast<float2 *>(orow + col0) = make_float2(o[0][r], o[1][r]);
                }
            }
            if (hi == 0) {
                omax[qrow] = vmax;
                osum[qrow] = vsum;
            }
        }
    };
    // this workgroup's triple: the rows of the result, or its slab of the split scratch
    auto store_mine = [&]() __attribute__((always_inline)) {
        const float my_max = T > 0 ? (m_ref + max_rel) * 0.69314718055994530942f : -INFINITY;
        if (a.kv_splits <= 1)
            store_rows(a.contrib, a.ldo, a.lmax, a.lsum, oacc, my_max, l_tot);
        else
            store_rows(a.ws_contrib + (size_t)split * a.ws_rows * a.ws_ld, a.ws_ld, a.ws_lmax + (size_t)split * a.ws_rows,
                       a.ws_lsum + (size_t)split * a.ws_rows, oacc, my_max, l_tot);
    };

    // (Q, the DMA offsets and the check-then-store order below are arranged so that the FIRST pass's hot
    //  loop is instruction for instruction the loop of the kernel without a second pass -- hipcc's
    //  allocation of these full-register-file kernels shifts with any liveness change around the loop;
    //  tests/test_kernel_isa.py holds it in place)
    do {                                      // SK: one trip per piece of this workgroup's run; otherwise one trip
    if constexpr (SK) {
        // (integer division runs on the VALU: readfirstlane tells hipcc the quotients are wave-uniform -- the
        //  DMA's base address must sit in SGPRs, and a "divergent" one costs a waterfall loop per piece)
        qblock = __builtin_amdgcn_readfirstlane(sk_pos / sk_ntiles);
        const int t0 = sk_pos - qblock * sk_ntiles;
        const int t1 = min(sk_ntiles, t0 + (sk_end - sk_pos));
        split = work - __builtin_amdgcn_readfirstlane((qblock * sk_ntiles) / kv_per_split);   // pieces of this query block before this one
        qrow = qblock * kQRowsPerBlock + wave * 32 + li;
        kv_begin = t0 * kKvTile;
        kv_end = min(a.n_local, t1 * kKvTile);
        T = t1 - t0;
        load_q();
    }
    run_pass();
#if SDPA_RANGE_REDO
    {   // range check of the deferred-rescale pass (see `defer` above).  The LDS tiles are dead here:
        // every wave has passed the last step's barrier behind its last fragment read.
        bool bad = !__builtin_isfinite(l_tot);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) bad |= !__builtin_isfinite(oacc[tt][r]);
        int *vote = reinterpret_cast<int *>(smem);
        const int wave_bad = __any(bad) ? 1 : 0;
        if (lane == 0) vote[wave] = wave_bad;
        __syncthreads();
        const int redo = __builtin_amdgcn_readfirstlane(vote[0] | vote[1] | vote[2] | vote[3]);
        if (redo) {
            __syncthreads();                  // the votes are read before the second pass's first DMA lands on them
            defer = 0.f;
            run_pass();
        }
    }
#endif
    store_mine();
    if constexpr (SK) {
        if (kv_begin + T * kKvTile >= sk_ntiles * kKvTile) {
            // the last piece of its query block: the slabs this block does not use get the empty triple
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[tt][r] = 0.f;
            for (int sp = split + 1; sp < a.kv_splits; ++sp)
                store_rows(a.ws_contrib + (size_t)sp * a.ws_rows * a.ws_ld, a.ws_ld, a.ws_lmax + (size_t)sp * a.ws_rows,
                           a.ws_lsum + (size_t)sp * a.ws_rows, oacc, -INFINITY, 0.f);
        }
        sk_pos += T;
        if (sk_pos < sk_end) __syncthreads();  // the range-check votes are read before the next piece's first DMA lands on them
    }
    } while (SK && sk_pos < sk_end);
    if (SK || a.kv_splits <= 1) return;
    if constexpr (!MERGE) {
        return;                               // the slots are merged by a later pass (split_merge_kernel)
    } else {

    // ---- in-kernel split merge: the LAST workgroup of this query block to arrive merges the block's
    // kv_splits partial triples (attention-mpi.c:340-351 applied inside one GPU).  Placement-independent
    // hand-off (the splits of a block run on different XCDs, whose L2s are not coherent): plain slab
    // stores, every wave drains them, one lane releases at agent scope and takes a ticket; the last
    // arriver acquires at agent scope and reads every slab -- its own included, in split order, so that
    // the sums do not depend on who arrived last (bitwise the separate merge pass's result).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *last_flag = reinterpret_cast<int *>(smem) + 8;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // restates the wait behind buffer_wbl2 where hipcc cannot drop it
        unsigned long long *word = a.tickets + qblock;
        unsigned long long seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned arrived;
        for (;;) {       // a word of another generation (or never written) counts as zero arrivals
            arrived = (seen >> 8) == a.ticket_tag ? (unsigned)(seen & 255u) : 0u;
            const unsigned long long next = (a.ticket_tag << 8) | (unsigned long long)(arrived + 1u);
            if (__hip_atomic_compare_exchange_strong(word, &seen, next, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT))
                break;
        }
        const int last = arrived + 1u == (unsigned)a.kv_splits;
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *last_flag = last;
    }
    __syncthreads();
    if (!__builtin_amdgcn_readfirstlane(*last_flag)) return;

    f32x16 macc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) macc[t][r] = 0.f;
    float gm = -INFINITY, tot = 0.f;
    if (qrow < a.m) {
        for (int sp = 0; sp < a.kv_splits; ++sp) gm = fmaxf(gm, a.ws_lmax[(size_t)sp * a.ws_rows + qrow]);
        for (int sp = 0; sp < a.kv_splits; ++sp) {
            const float lm = a.ws_lmax[(size_t)sp * a.ws_rows + qrow];
            const float w = (lm == -INFINITY) ? 0.f : expf(lm - gm);
            tot = fmaf(w, a.ws_lsum[(size_t)sp * a.ws_rows + qrow], tot);
            const float *srow = a.ws_contrib + ((size_t)sp * a.ws_rows + qrow) * a.ws_ld;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col0 = NV * crow(r, hi);
                if constexpr (NV == 4) {
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const float4 o = *reinterpret_cast<const float4 *>(srow + 128 * h + col0);
                        macc[4 * h][r] = fmaf(w, o.x, macc[4 * h][r]);
                        macc[4 * h + 1][r] = fmaf(w, o.y, macc[4 * h + 1][r]);
                        macc[4 * h + 2][r] = fmaf(w, o.z, macc[4 * h + 2][r]);
                        macc[4 * h + 3][r] = fmaf(w, o.w, macc[4 * h + 3][r]);
                    }
                } else {
                    const float2 o = *reinterpret_cast<const float2 *>(srow + col0);
                    macc[0][r] = fmaf(w, o.x, macc[0][r]);
                    macc[1][r] = fmaf(w, o.y, macc[1][r]);
                }
            }
        }
    }
    store_rows(a.contrib, a.ldo, a.lmax, a.lsum, macc, gm, tot);
    }   // MERGE
}

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

// ---------------------------------------------------------------------------
// Any-shape kernel (dk or dv > 128): one wave per query row, lanes across key
// rows for the dot products and across value columns for the accumulate.  A
// correctness path for shapes the MFMA kernel does not cover in fp32; slow.
// ---------------------------------------------------------------------------
constexpr int kGenericMaxCols = 16;   // dv <= 64 * 16
static const int kGenericMaxColsAnchor = 0;   // (an address of this library's own image: see next_ticket_tag)

__global__ __launch_bounds__(256) void generic_partial_kernel(PartialArgs a, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    float *qs = smem + (size_t)wave * a.ldq;
    if (row < a.m)
        for (int t = lane; t < a.ldq; t += 64) qs[t] = a.Q[(size_t)row * a.ldq + t];
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

    // ---- stream-K instead?  (pipelined kernels only; $SDPA_STREAMK = 0 / 1 / auto)
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
    int s = 1;
    for (int cus = kChipCus; cus >= kChipCus / 2; cus -= 8) s = std::max(s, pick_kv_splits(m, n_local, dk, dv, cus));
    return workspace_bytes_for(m, dv, s);
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
    // also merge them itself ($SDPA_SPLIT_MERGE=kernel: the last workgroup of a query
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
            if ((e = attr.ensure(&fused_pipelined_kernel<DK, DV, 0, 0, 1>, dev, lds)) != hipSuccess) return e;
            hipLaunchKernelGGL((fused_pipelined_kernel<DK, DV, 0, 0, 1>), dim3(plan.workers), dim3(256), lds, s, k,
                               plan.run, nqb, scale);
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
    } else {
        static AttrOnce attr;
        if ((e = attr.ensure(&fused_pipelined_kernel<DK, DV, ABL, 0>, dev, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((fused_pipelined_kernel<DK, DV, ABL, 0>), dim3(nqb * k.kv_splits), dim3(256), lds, s,
                           k, kv_per_split, nqb, scale);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (k.kv_splits > 1 && !k.defer_merge && !k.tickets) e = launch_split_merge(k, s);
    return e;
}

hipError_t launch_shard_partial(const PartialArgs &a_in, hipStream_t s) {
    PartialArgs a = a_in;
    if (a.ws_rows <= 0) a.ws_rows = a.m;
#ifdef SDPA_ABLATIONS   // tools/ builds only: the shipped library never reads $SDPA_TUNE
    static const int tune_env = getenv("SDPA_TUNE") ? atoi(getenv("SDPA_TUNE")) : 0;
    a.tune = tune_env;
#else
    a.tune = 0;
#endif
    // Operand images whose rows are 64 / 128 / 256 floats wide (dims padded with zero columns, the
    // header's contract for columns [dk, ld)) take the software-pipelined LDS-DMA kernel of that width:
    // same padded MFMA work as the any-shape kernels, at the pipelined kernel's rate.  The softmax
    // scale is 1/sqrt of the TRUE dk; contrib rows must hold the padded width ($SDPA_TUNE&4: off).
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
    if (uses_dksplit(a.dk, a.dv) && !(a.tune & 8))     // $SDPA_TUNE&8: the kernels it replaced
        return launch_dksplit(a, s);                    // sdpa_fwd_f32_dksplit.hip
    if (a.dk > kMaxMfmaDk) {
        const size_t lds = (size_t)4 * a.ldq * sizeof(float);
        if (lds > 64 * 1024) return hipErrorInvalidValue;          // dk <= 4096
        const float scale = 1.0f / sqrtf((float)a.dk);
        // the kernel holds 64 * kGenericMaxCols value columns per row: wider V goes in column chunks, one
        // launch each (the scores are recomputed; lmax / lsum come out the same from every chunk)
        for (int c0 = 0; c0 < a.dv; c0 += 64 * kGenericMaxCols) {
            PartialArgs ac = a;
            ac.V = a.V + c0;
            ac.contrib = a.contrib + c0;
            ac.dv = std::min(64 * kGenericMaxCols, a.dv - c0);
            hipLaunchKernelGGL(generic_partial_kernel, dim3((a.m + 3) / 4), dim3(256), lds, s, ac, scale);
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

}  // namespace sdpa
