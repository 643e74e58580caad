// mfma_probe.hip -- what does v_mfma_f32_32x32x16_bf16 sustain on this part, as a function of
// accumulator-chain shape and waves per SIMD?  Sizes the d=512 bf16 kernel's expectations:
//   chain<1>  every MFMA accumulates into the same 16 registers (the QK^T phase of one wave)
//   chain<2>  two alternating accumulators
//   chain<8>  eight independent accumulators (the P.V phase)
//   mixed     the d=512 step shape: 32 dependent + 16 over 8 accumulators, with N VALU ops between
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void chain(float *out, int iters, unsigned seed) {
    u32x4 a = {seed + threadIdx.x, seed * 3u, 0x3f803f80u, 0x3f803f80u};
    u32x4 b = {0x3f803f80u, seed, 0x3f803f80u, threadIdx.x};
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k)
            acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                                    __builtin_bit_cast(bf16x8, b), acc[k % NACC], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}


// same as chain<NACC> but with operands that toggle: random bf16 bit patterns, four different A
// and B fragments in rotation (a power/clock question, not an issue-rate one)
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
template <int NACC>
__global__ __launch_bounds__(256) void chain_rand(float *out, int iters, unsigned seed) {
    u32x4 av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h = hash32(seed + threadIdx.x * 9781u + blockIdx.x * 6271u + i * 131u);
        // keep exponents moderate so nothing overflows: sign + random mantissa, exponent ~ 2^-4..2^3
        auto mk = [&](unsigned r) { return (r & 0x807f807fu) | 0x3d803d80u | ((r >> 3) & 0x03000300u); };
        av[i] = u32x4{mk(h), mk(hash32(h + 1)), mk(hash32(h + 2)), mk(hash32(h + 3))};
        bv[i] = u32x4{mk(hash32(h + 4)), mk(hash32(h + 5)), mk(hash32(h + 6)), mk(hash32(h + 7))};
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k)
            acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[k % 4]),
                                                                    __builtin_bit_cast(bf16x8, bv[(k / 4) % 4]), acc[k % NACC], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

// step shape of the d=512 kernel: 32-long dependent chain with NV VALU ops after every 2nd MFMA,
// then 16 MFMAs over 8 accumulators
template <int NV>
__global__ __launch_bounds__(256) void mixed(float *out, int iters, unsigned seed) {
    u32x4 a = {seed + threadIdx.x, seed * 3u, 0x3f803f80u, 0x3f803f80u};
    u32x4 b = {0x3f803f80u, seed, 0x3f803f80u, threadIdx.x};
    f32x16 o[8], s;
    float v = (float)threadIdx.x, w = 1.0001f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), s, 0, 0, 0);
            if (k & 1) {
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(w));
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            o[k % 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), o[k % 8], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV / 2; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(w));
        }
        __builtin_amdgcn_sched_barrier(0);
        b.x ^= __builtin_bit_cast(unsigned, s[0]) & 1u;      // the chain result is consumed
    }
    float t = v;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += o[i][r];
    if (t == 12345.678f) out[0] = t;
}


// OP: 0 dependent fma chain, 1 independent fmas (8 registers round-robin), 2 independent v_exp_f32,
//     3 v_accvgpr_read of the score tile, 4 independent fma + one ds_read_b128 per MFMA
// PER: VALU ops after EVERY MFMA of the dependent chain (and after every P.V MFMA)
template <int OP, int PER>
__global__ __launch_bounds__(256) void mixed2(float *out, int iters, unsigned seed) {
    __shared__ u32x4 lds[1024];
    u32x4 a = {seed + threadIdx.x, seed * 3u, 0x3f803f80u, 0x3f803f80u};
    u32x4 b = {0x3f803f80u, seed, 0x3f803f80u, threadIdx.x};
    lds[threadIdx.x] = a; lds[threadIdx.x + 256] = b;
    __syncthreads();
    f32x16 o[8], s;
    float v[8];
    const float w = 1.0001f;
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    f32x2_ pv[4];
    const f32x2_ pw = {1.0001f, 0.9999f};
#pragma unroll
    for (int j = 0; j < 4; ++j) pv[j] = f32x2_{(float)j, (float)threadIdx.x};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)(threadIdx.x + j);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    u32x4 frag = a;
    int n = 0;
    auto valu = [&](int cnt) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < cnt; ++j, ++n) {
            if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[0]) : "v"(w));
            if constexpr (OP == 1 || OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[n % 8]) : "v"(w));
            if constexpr (OP == 2) asm volatile("v_exp_f32 %0, %0\n\ts_nop 0" : "+v"(v[n % 8]));
            if constexpr (OP == 3) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[n % 8]) : "a"(s[n % 16]));
            if constexpr (OP == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(v[n % 8]));
            if constexpr (OP == 7) asm volatile("v_exp_f16 %0, %0" : "+v"(v[n % 8]));
            if constexpr (OP == 8) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(v[n % 8]));
            if constexpr (OP == 9) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[n % 8]) : "v"(w));
            if constexpr (OP == 10) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[n % 8]) : "v"(w));
            if constexpr (OP == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pv[n % 4]) : "v"(pw));
            if constexpr (OP == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[n % 8]) : "s"(seed));
            if constexpr (OP == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[n % 8]));
        }
    };
    for (int it = 0; it < iters; ++it) {
        n = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        if constexpr (OP == 3) asm volatile("" : "+a"(s));
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, frag), __builtin_bit_cast(bf16x8, b), s, 0, 0, 0);
            if constexpr (OP == 4) frag = lds[(threadIdx.x + 64 * k) & 1023];
            if constexpr (OP == 3) { if (k < 31) continue; }
            valu(PER);
        }
        if constexpr (OP == 3) { __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            o[k % 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, frag), __builtin_bit_cast(bf16x8, b), o[k % 8], 0, 0, 0);
            if constexpr (OP == 4) frag = lds[(threadIdx.x + 64 * k + 7) & 1023];
            valu(OP == 3 ? 2 * PER : PER);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (OP != 3) b.x ^= __builtin_bit_cast(unsigned, s[0]) & 1u;
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += v[j] + pv[j % 4][0] + pv[j % 4][1];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += o[i][r];
    if (t == 12345.678f) out[0] = t;
}


// fp32: does VALU work overlap v_mfma_f32_32x32x2_f32 (64 cycles each) on the same SIMD?  The part's
// vector-fp32 and matrix-fp32 peaks are the same number, which suggests they share the FMA lanes.
// OP: 1 independent v_fma_f32, 2 independent v_exp_f32, 5 v_pk_fma_f32 (two fp32 per lane and op)
template <int OP, int PER>
__global__ __launch_bounds__(256) void f32mix(float *out, int iters, unsigned seed) {
    float a = 1.0f + 1e-3f * threadIdx.x, b = 1.0f - 1e-3f * (seed & 7);
    f32x16 o[4];
    float v[8];
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 pv[8];
    const float w = 1.0001f;
    const f32x2 pw = {1.0001f, 0.9999f};
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = (float)(threadIdx.x + j); pv[j] = f32x2{v[j], v[j] + 1.f}; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            o[k % 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, o[k % 4], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                constexpr int dummy = 0; (void)dummy;
                const int q = (k * PER + j) % 8;      // compile-time after unrolling
                if constexpr (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q]) : "v"(w));
                if constexpr (OP == 2) asm volatile("v_exp_f32 %0, %0\n\ts_nop 0" : "+v"(v[q]));
                if constexpr (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[q]) : "v"(pw));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += v[j] + pv[j][0] + pv[j][1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += o[i][r];
    if (t == 12345.678f) out[0] = t;
}

template <typename F>
static void run_f32(const char *name, F launch, int blocks, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(blocks, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(blocks, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfmas = (double)blocks * 4 * iters * 32.0;
    const double tf = mfmas * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-28s blocks=%5d  %8.3f ms  %8.1f TFLOP/s  (%.1f%% of 157.3)\n", name, blocks, ms, tf, 100 * tf / 157.3);
}
#define F32(OP, PER) run_f32("f32mix<op" #OP ",per" #PER ">", [&](int g, int n) { hipLaunchKernelGGL((f32mix<OP, PER>), dim3(g), dim3(256), 0, 0, out, n, 1u); }, blocks, 2000)

template <typename F>
static void run(const char *name, F launch, double mfma_per_thread_block_iter, int blocks, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(blocks, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(blocks, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfmas = (double)blocks * 4 /*waves*/ * iters * mfma_per_thread_block_iter;
    const double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD at a nominal 2.4 GHz, given waves share SIMDs evenly
    printf("%-28s blocks=%5d  %8.3f ms  %8.1f TFLOP/s  (%.1f%% of 2516)\n", name, blocks, ms, tf, 100 * tf / 2516.0);
}

#define M2(OP, PER) run("mixed2<op" #OP ",per" #PER ">", [&](int g, int n) { hipLaunchKernelGGL((mixed2<OP, PER>), dim3(g), dim3(256), 0, 0, out, n, 1u); }, 48, blocks, it)
int main(int argc, char **argv) {
    float *out; CK(hipMalloc(&out, 64));
    const int it = 3000;
    const int maxw = argc > 1 ? atoi(argv[1]) : 1;
    for (int bpc = 1; bpc <= maxw; ++bpc) {
        const int blocks = 256 * bpc;
        printf("-- %d wave(s) per SIMD\n", bpc);
        run("chain<1>", [&](int g, int n) { hipLaunchKernelGGL(chain<1>, dim3(g), dim3(256), 0, 0, out, n, 1u); }, 64, blocks, it);
        run("chain<8>", [&](int g, int n) { hipLaunchKernelGGL(chain<8>, dim3(g), dim3(256), 0, 0, out, n, 1u); }, 64, blocks, it);
        run("chain_rand<1> (toggling data)", [&](int g, int n) { hipLaunchKernelGGL(chain_rand<1>, dim3(g), dim3(256), 0, 0, out, n, 7u); }, 64, blocks, it);
        run("chain_rand<8> (toggling data)", [&](int g, int n) { hipLaunchKernelGGL(chain_rand<8>, dim3(g), dim3(256), 0, 0, out, n, 7u); }, 64, blocks, it);
        run("chain_rand<8> x4 longer", [&](int g, int n) { hipLaunchKernelGGL(chain_rand<8>, dim3(g), dim3(256), 0, 0, out, n * 4, 7u); }, 64 * 4, blocks, it);
        if (argc > 2 && atoi(argv[2]) == 2) {      // fp32 MFMA + VALU overlap question only
            F32(1, 0); F32(1, 2); F32(1, 4); F32(1, 8); F32(1, 12); F32(1, 16);
            F32(2, 1); F32(2, 2); F32(2, 4);
            F32(5, 4); F32(5, 8);
            continue;
        }
        if (argc > 2 && atoi(argv[2]) == 3) {      // cost of single VALU ops beside bf16 MFMAs
            M2(1, 4); M2(1, 8);
            M2(6, 2); M2(6, 4); M2(7, 2); M2(7, 4); M2(8, 2); M2(8, 4); M2(13, 2); M2(13, 4);
            M2(9, 4); M2(9, 8); M2(10, 4); M2(10, 8); M2(11, 4); M2(11, 8); M2(12, 4); M2(12, 8);
            continue;
        }
        if (argc > 2) continue;
        run("mixed<8> (dep, every 2nd)", [&](int g, int n) { hipLaunchKernelGGL(mixed<8>, dim3(g), dim3(256), 0, 0, out, n, 1u); }, 48, blocks, it);
        M2(0, 2); M2(0, 4); M2(0, 6);
        M2(1, 2); M2(1, 4); M2(1, 6); M2(1, 8); M2(1, 10);
        M2(2, 1); M2(2, 2); M2(2, 3); M2(2, 4);

        M2(4, 0); M2(4, 2); M2(4, 4); M2(4, 6);
    }
    return 0;
}
