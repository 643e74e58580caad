// sdpa_internal.h -- launchers shared between the kernel translation units and
// the C-ABI layer.  Not part of the public interface (include/sdpa_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace sdpa {

constexpr int kQRowsPerBlock = 128;   // 4 waves x 32 query rows
constexpr int kKvTile        = 32;    // K/V rows per LDS tile
constexpr int kMaxFastDim    = 128;   // dk, dv <= 128 take the MFMA kernel

struct PartialArgs {
    const float *Q;  int ldq;
    const float *K;  int ldk;
    const float *V;  int ldv;
    float *contrib;  int ldo;      // [m x dv] un-normalised
    float *lmax;                   // [m]
    float *lsum;                   // [m]
    int m, n_local, dk, dv;
    int kv_splits;                 // >= 1
    float *ws_contrib;             // [kv_splits x m x ws_ld] (kv_splits > 1 only)
    int ws_ld;                     // row stride of ws_contrib = dv rounded up to 4
    float *ws_lmax;                // [kv_splits x m]
    float *ws_lsum;                // [kv_splits x m]
    int tune;                      // experiment switches ($SDPA_TUNE), 0 = shipped default
};

int  pick_kv_splits(int m, int n_local, int dk, int dv);
size_t workspace_bytes(int m, int n_local, int dk, int dv);

// Enqueue the fused kernel (+ the split merge when kv_splits > 1).
hipError_t launch_shard_partial(const PartialArgs &a, hipStream_t s);

hipError_t launch_cvt_d2f(const double *src, float *dst, long rows, int cols, int ld, hipStream_t s);
hipError_t launch_cvt_f2d(const float *src, int ld, double *dst, long rows, int cols, hipStream_t s);
hipError_t launch_merge_rescale(float *contrib, int ldo, float *lsum, const float *lmax,
                                const float *gmax, int m, int dv, hipStream_t s);
hipError_t launch_merge_normalise(float *contrib, int ldo, const float *gsum, int m, int dv,
                                  hipStream_t s);
hipError_t launch_finish_f64(const float *contrib, int ldo, const float *lsum, double *result,
                             int m, int dv, hipStream_t s);

}  // namespace sdpa
