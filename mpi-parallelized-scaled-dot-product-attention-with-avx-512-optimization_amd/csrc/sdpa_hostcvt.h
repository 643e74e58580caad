// sdpa_hostcvt.h -- fp64 -> operand-image conversion on HOST threads (internal; $SDPA_HOST_CVT=1).
//
// The reference converts K, V and every Q batch on the host before they travel
// (cvt_d2f_avx512, attention-mpi.c:31-64, called at :224-225, :248-249, :303, :325).  The default
// here is the opposite -- fp64 crosses PCIe and the device converts -- because one rank's 8 MPI-era
// cores are no match for HBM.  But a boundary call on short problems is bound by the PCIe bytes, and
// the GPU box's host has a hundred cores: this is the reference's own placement, multi-threaded,
// writing the operand images (fp32: half the bytes, bf16: a quarter) into page-locked staging that
// goes over the link as it is.  Bit for bit the device converters' results (same roundings).
#pragma once
#include <stddef.h>

namespace sdpa {

enum CvtKind {
    kCvtF32 = 0,      // float image, rows padded with zero columns to ld floats (cvt_d2f_kernel)
    kCvtBf16 = 1,     // bf16 image, rows padded to ld, values bf16((float)(x * mult)) (cvt_d2bf_kernel)
    kCvtBf16T = 2,    // the TRANSPOSED bf16 image of V (cvt_d2bf_t_kernel): submit_t() / host_convert_vt() only
    kCvtBf16Swz = 3,  // rows of the TILED K image (dv > 256, sdpa_internal.h): kCvtBf16's values, 16-byte chunk c of image row r at
                      // c ^ (r & swz), swz = min(15, ld / 8 - 1); dst must be an image row that is a multiple of 16
};

// Streaming (non-temporal) stores in the AVX-512 rows: whole 64-byte lines written past the cache wherever a row's
// destination is line-aligned (the staging images are).  $SDPA_DEBUG host_cvt_nt overrides this default.
#ifndef SDPA_HOST_CVT_NT_DEFAULT
#define SDPA_HOST_CVT_NT_DEFAULT 1
#endif

// `rows` rows on the calling thread (AVX-512 where the CPU has it; force_scalar: the plain-C rows;
// stream_stores: 1 / 0 = streaming stores on / off, -1 = the default)
void host_convert_rows(const double *src, void *dst, long rows, int cols, int ld, CvtKind kind, double mult,
                       bool force_scalar, int stream_stores = -1);

// cols > 256: the TILED Vt image (sdpa_internal.h) -- dst = the block of the first key's tile, whole tiles, ldt unused.  Otherwise:
// `keys` rows of V (fp64, `cols` columns) -> the columns [0, keys_pad) of a Vt image whose rows are `ldt` elements apart (dst = its
// row 0 at the first key's tile; keys_pad = keys rounded up to whole 32-key tiles, or more: zero tiles): dst[c * ldt + kvpos(j)] for
// key j (sdpa_internal.h: bf16_kvpos), zero behind the last key, zero rows [cols, cols_pad) -- cvt_d2bf_t_kernel's image bit for bit
void host_convert_vt(const double *src, unsigned short *dst, long keys, long keys_pad, int cols, int cols_pad, long ldt,
                     bool force_scalar, int stream_stores = -1);

// dst[i] = (double)src[i] on the calling thread (cvt_f2d_avx512, attention-mpi.c:68-101)
void host_widen(const float *src, double *dst, size_t n, bool force_scalar);

class HostConverter {
public:
    static HostConverter *create(int threads);      // nullptr on failure
    virtual ~HostConverter() {}
    virtual int threads() const = 0;
    // page-locked staging area `which` (0 = K image, 1 = V image, 2 = Q image, 3 = fp32 result rows), grown to `bytes`
    virtual void *staging(int which, size_t bytes) = 0;
    // one call: begin(), submit() every conversion in the order the copies will need them, kick();
    // wait(task) in front of the copy that reads the task's output; finish() before the caller's arrays
    // are handed back (also on error paths)
    virtual void begin() = 0;
    virtual int submit(const double *src, void *dst, long rows, int cols, int ld, CvtKind kind, double mult) = 0;
    // the same for a column range of a Vt image (host_convert_vt's arguments; whole 32-key tiles per work item)
    virtual int submit_t(const double *src, unsigned short *dst, long keys, long keys_pad, int cols, int cols_pad, long ldt) = 0;
    // before kick(): confine the pool's threads to the NUMA node the call's source arrays live on (sampled pages; all CPUs again
    // when the samples disagree or the host says nothing) -- only with $SDPA_DEBUG host_cvt_pin=1 (opt-in: it did not pay, sdpa_hostcvt.cpp).
    // placed_node(): where they are (-1: anywhere)
    virtual void place_near(const void *const *arrays, const size_t *bytes, int n_arrays) = 0;
    virtual int placed_node() const = 0;
    virtual void kick() = 0;
    virtual void wait(int task) = 0;
    virtual void finish() = 0;
    // fp32 result rows -> the caller's fp64 array, on the pool's threads and the calling one; returns when done.
    // Independent of begin()/finish(): may be called while a batch of input conversions is still being worked on.
    virtual void widen(const float *src, double *dst, size_t n) = 0;
};

}  // namespace sdpa
