// sdpa_f32_device.h -- device helpers shared by the fp32 kernel translation units (internal).
#pragma once
#include "sdpa_internal.h"

#include <math.h>
#include <stdint.h>

namespace sdpa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x1 __attribute__((ext_vector_type(1)));

// key row (within a 32-row tile) held in accumulator register r of half-wave hi
// for the 32x32 MFMA C/D layout: row = (r&3) + 8*(r>>2) + 4*hi.
__device__ __forceinline__ constexpr int crow(int r, int hi) {
    return (r & 3) + 8 * (r >> 2) + 4 * hi;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// exp2 / max as volatile asm: ordered against sched_barrier() and each other, so they stay in
// the MFMA shadow they were written in (the compiler otherwise gathers pure VALU ops after the
// MFMA block).  s_nop 0: a TRANS result needs one wait state before a non-TRANS VALU reads it.
__device__ __forceinline__ float pinned_exp2(float x) {
    float y;
    asm volatile("v_exp_f32 %0, %1\n\ts_nop 0" : "=v"(y) : "v"(x));
    return y;
}
__device__ __forceinline__ float pinned_max3(float a, float b, float c) {
    float y;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(y) : "v"(a), "v"(b), "v"(c));
    return y;
}
__device__ __forceinline__ float pinned_max(float a, float b) {
    float y;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b));
    return y;
}

__device__ __forceinline__ float pinned_add(float a, float b) {
    float y;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b));
    return y;
}

// bijective "contiguous chunk per XCD" remap of a 1-D grid: hardware places
// block b on XCD b%8; give each XCD a contiguous range of work items so blocks
// that share a K/V split share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int total) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = total >> 3, r = total & 7;
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return first + slot;
}

// NT consecutive floats of one V row (one per 32-column O^T tile) as a native vector.
template <int NT> struct VFrag;
template <> struct VFrag<4> {
    f32x4 v;
    static __device__ __forceinline__ VFrag load(const float *p) { return {*reinterpret_cast<const f32x4 *>(p)}; }
};
template <> struct VFrag<8> {      // two float4, 128 columns apart (tiles 0..3 and 4..7)
    f32x8 v;
    static __device__ __forceinline__ VFrag load(const float *p) {
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(p), up = *reinterpret_cast<const f32x4 *>(p + 128);
        return {__builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7)};
    }
};
template <> struct VFrag<2> {
    f32x2 v;
    static __device__ __forceinline__ VFrag load(const float *p) { return {*reinterpret_cast<const f32x2 *>(p)}; }
};
template <> struct VFrag<1> {
    f32x1 v;
    static __device__ __forceinline__ VFrag load(const float *p) { VFrag f; f.v[0] = *p; return f; }
};

// NT CONSECUTIVE floats of one V row (the dk-split kernels: O^T tile tt of a wave's slice holds column
// NT * li + tt), 4-byte aligned; 3 / 6 / 8 floats come as dwordx3 / dwordx4 + dwordx2 / 2 x dwordx4.
template <int NT> struct VRun {
    typedef float vec_t __attribute__((ext_vector_type(NT)));
    vec_t v;
    static __device__ __forceinline__ VRun load(const float *p) {
        VRun f;
        if constexpr (NT == 1) {
            f.v[0] = *p;
        } else if constexpr (NT == 2 || NT == 4) {
            f.v = *reinterpret_cast<const vec_t *>(p);       // 8- / 16-byte aligned by construction (NT * li floats into an aligned row)
        } else {
            float t[NT];
            __builtin_memcpy(t, p, NT * sizeof(float));
#pragma unroll
            for (int i = 0; i < NT; ++i) f.v[i] = t[i];
        }
        return f;
    }
};

}  // namespace sdpa
