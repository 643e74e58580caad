// sdpa_errors.h -- error plumbing shared by the C-ABI translation units (internal).
#pragma once
#include "sdpa_debug.h"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/sdpa_hip.h"
#include "sdpa_internal.h"

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "sdpa: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                   \
            return e_ == hipErrorOutOfMemory ? SDPA_ENOMEM : SDPA_EHIP;                    \
        }                                                                                  \
    } while (0)

#define SDPA_TRY(expr)                \
    do {                              \
        int c_ = (expr);              \
        if (c_ != SDPA_OK) return c_; \
    } while (0)

namespace sdpa {

inline int round4(int x) { return (x + 3) / 4 * 4; }

// A compute stream whose fused kernels leave `reserve_cus` compute units' worth of workgroup slots (rounded up to a
// multiple of the XCD count) to the kernels of OTHER streams -- RCCL's, the merge kernels, the converts.
// reserve_cus <= 0: an ordinary non-blocking stream.
//
// Round 4: the reservation is made by GRID SIZE, not by a CU mask.  The stream is an ordinary non-blocking
// stream registered with (CUs - reserve) compute units; the fp32 pipelined launchers size their stream-K grid by
// that figure (2 workgroups per CU), so 2 x reserve of the chip's 512 workgroup slots stay empty -- on CUs that
// hold ONE fused workgroup and therefore have half their registers and 96 KiB of LDS free for a co-resident
// kernel.  Why not hipExtStreamCreateWithCUMask (round 3): measured on MI355X (profiles/r04/streamk_ab.log), a
// masked stream runs the exactly-fitting stream-K grid at 0.60 of the unmasked rate -- the mask takes one CU out
// of ONE shader engine per XCD while the dispatcher keeps dealing workgroups to the engines evenly, so a few
// workgroups per XCD queue for a second round; the masked stream is also a BLOCKING stream (it synchronises
// with the legacy NULL stream) and its bit -> XCD layout depends on the partition mode.
// $SDPA_DEBUG reserve_by_mask=1 keeps the masked form for A/B runs.  Kernels without a stream-K form (dk-split, bf16,
// 256-wide fp32) launch their full grids on such a stream as on any other.
inline int create_masked_stream(hipStream_t *out, int reserve_cus) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount, xcds = cus >= 64 ? cus / 32 : 1;
    if (reserve_cus <= 0 || cus < 2 * xcds || cus > 1024) {
        HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
        return SDPA_OK;
    }
    int r = (reserve_cus + xcds - 1) / xcds * xcds;
    if (r > cus / 2) r = cus / 2 / xcds * xcds;
    if (sdpa_debug_int("reserve_by_mask", 0) != 0) {
        // (the mask's bits are dealt round-robin over the XCDs, bit i -> XCD i % xcds: the same number from each)
        unsigned mask[32] = {0};
        for (int i = 0; i < cus - r; ++i) mask[i / 32] |= 1u << (i % 32);
        HIP_TRY(hipExtStreamCreateWithCUMask(out, (unsigned)((cus + 31) / 32), mask));
    } else {
        HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    }
    register_stream_cus(*out, cus - r);      // the fused launchers size their stream-K grids by it
    return SDPA_OK;
}

inline int require_device() {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
        (void)hipGetLastError();
        return SDPA_ENODEV;
    }
    return SDPA_OK;
}

}  // namespace sdpa
