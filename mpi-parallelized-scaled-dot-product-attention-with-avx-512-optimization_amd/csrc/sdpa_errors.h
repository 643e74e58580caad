// sdpa_errors.h -- error plumbing shared by the C-ABI translation units (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/sdpa_hip.h"

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

inline int require_device() {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
        (void)hipGetLastError();
        return SDPA_ENODEV;
    }
    return SDPA_OK;
}

}  // namespace sdpa
