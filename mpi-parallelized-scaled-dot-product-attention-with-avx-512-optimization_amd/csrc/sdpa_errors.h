// sdpa_errors.h -- error plumbing shared by the C-ABI translation units (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

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

// A stream whose kernels may use all but `reserve_cus` compute units of the current device (rounded up
// to a multiple of the XCD count, the same number taken from every XCD: the mask's bits are dealt
// round-robin over the XCDs, bit i -> XCD i % xcds).  reserve_cus <= 0: an ordinary non-blocking stream.
// Two things a caller must know: (1) hipExtStreamCreateWithCUMask has no flags argument and
// creates a BLOCKING stream -- it synchronises implicitly with the legacy NULL stream, so nothing may be
// enqueued on the NULL stream (a synchronous hipMemcpy, PyTorch's default stream) while the overlap it
// exists for is wanted; the C host uses only its own non-blocking streams beside it.  (2) The XCD count is
// the whole-chip SPX figure, 8 for the 256-CU MI355X, derived as CUs / 32; in a CPX/partitioned mode a
// device is one XCD (32 CUs) and the mask degenerates to "the last r CUs", which is still a valid mask.
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
    unsigned mask[32] = {0};
    for (int i = 0; i < cus - r; ++i) mask[i / 32] |= 1u << (i % 32);
    HIP_TRY(hipExtStreamCreateWithCUMask(out, (unsigned)((cus + 31) / 32), mask));
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
