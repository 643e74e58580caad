// sdpa_api.hip -- the C ABI declared in include/sdpa_hip.h: device level and the small
// stateless entry points.  The host level (sdpa_init / sdpa_attention_f64 / sdpa_prepare,
// the Q-batch pipeline and the collectives) lives in sdpa_host.hip and sdpa_coll.hip.
//
// Device level: thin argument-checking wrappers over the launchers of sdpa_fwd_f32.hip,
// sdpa_fwd_bf16.hip and sdpa_aux.hip, for hosts that own device memory and collectives
// themselves (one process per GPU with RCCL through torch.distributed).
#include "sdpa_errors.h"
#include "sdpa_hostcvt.h"
#include "sdpa_internal.h"
#include "sdpa_debug.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

namespace sdpa {

// ---- CU budget of the streams this library created with a reservation -------------------------------------------
namespace {
std::mutex &cus_mu() { static std::mutex *m = new std::mutex; return *m; }
std::vector<std::pair<hipStream_t, int>> &cus_tab() { static auto *t = new std::vector<std::pair<hipStream_t, int>>; return *t; }
std::atomic<int> dev_cus[64];
}  // namespace

void register_stream_cus(hipStream_t s, int cus) {
    std::lock_guard<std::mutex> lk(cus_mu());
    for (auto &e : cus_tab())
        if (e.first == s) { e.second = cus; return; }
    cus_tab().push_back({s, cus});
}

void forget_stream_cus(hipStream_t s) {
    std::lock_guard<std::mutex> lk(cus_mu());
    auto &t = cus_tab();
    for (size_t i = 0; i < t.size(); ++i)
        if (t[i].first == s) { t[i] = t.back(); t.pop_back(); return; }
}

int stream_cus(hipStream_t s) {
    {
        std::lock_guard<std::mutex> lk(cus_mu());
        for (auto &e : cus_tab())
            if (e.first == s) return e.second;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return kChipCus; }
    int c = dev_cus[dev].load(std::memory_order_relaxed);
    if (c <= 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) {
            (void)hipGetLastError();
            c = kChipCus;
        }
        dev_cus[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}

// ---- launch-path knobs: one immutable snapshot of the environment -------------------------------------------
namespace {
std::atomic<const LaunchKnobs *> knobs_now{nullptr};
const LaunchKnobs *read_knobs() {
    LaunchKnobs *k = new LaunchKnobs;      // (snapshots are never freed: a launcher on another thread may still hold one)
    k->split_merge_kernel = sdpa_debug_is("split_merge", "kernel") ? 1 : 0;
    k->streamk = (!sdpa_debug_find("streamk") || sdpa_debug_is("streamk", "auto")) ? -1 : (sdpa_debug_int("streamk", 0) != 0 ? 1 : 0);
    return k;
}
}  // namespace

const LaunchKnobs &launch_knobs() {
    const LaunchKnobs *k = knobs_now.load(std::memory_order_acquire);
    if (!k) {
        static std::mutex *mu = new std::mutex;
        std::lock_guard<std::mutex> lk(*mu);
        k = knobs_now.load(std::memory_order_acquire);
        if (!k) {
            k = read_knobs();
            knobs_now.store(k, std::memory_order_release);
        }
    }
    return *k;
}

void reload_launch_knobs() { knobs_now.store(read_knobs(), std::memory_order_release); }

// ---- the calling thread's last fused launch ----------------------------------------------------------------
namespace {
thread_local LaunchNote tl_note = {};
}
void note_launch(const char *kernel, int ntarg, int t0, int t1, int t2, int t3, int t4, int grid, int splits, int streamk,
                 int rows, int keys) {
    LaunchNote &n = tl_note;
    n.kernel = kernel;
    n.ntarg = ntarg;
    n.targ[0] = t0; n.targ[1] = t1; n.targ[2] = t2; n.targ[3] = t3; n.targ[4] = t4;
    n.grid = grid; n.splits = splits; n.streamk = streamk; n.rows = rows; n.keys = keys;
}
const LaunchNote &last_launch_note() { return tl_note; }
void set_launch_note(const LaunchNote &n) { tl_note = n; }
int format_launch_kernel(const LaunchNote &n, char *buf, size_t len) {
    if (!n.kernel) return snprintf(buf, len, "%s", "");
    char args[64];
    int at = 0;
    for (int i = 0; i < n.ntarg && i < 5; ++i) at += snprintf(args + at, sizeof args - at, "%s%d", i ? "," : "", n.targ[i]);
    args[at] = 0;
    return snprintf(buf, len, "sdpa::%s<%s>", n.kernel, args);
}

}  // namespace sdpa

namespace {

using sdpa::PartialArgs;
using sdpa::require_device;
using sdpa::round4;

int check_ld(int ld, int cols) { return (ld >= cols && ld % 4 == 0) ? SDPA_OK : SDPA_EINVAL; }

bool misaligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace

extern "C" {

#ifndef SDPA_BUILD_STAMP
#define SDPA_BUILD_STAMP "hipcc unknown; src unknown"
#endif
const char *sdpa_version(void) { return "sdpa-hip 0.6 abi 6 (gfx950, f32 + bf16 MFMA; " SDPA_BUILD_STAMP ")"; }
int sdpa_abi_version(void) { return SDPA_ABI_VERSION; }

void sdpa_reload_env(void) { sdpa::reload_launch_knobs(); }

// audit builds (-DSDPA_DMA_ASSERT on the kernel translation units, tools/build_variant.sh): out[0] = source addresses
// of LDS-DMA pieces / clamped fragment loads found outside their operand image, out[1] = audited launches; both 0
// in the shipped library.  Not in include/sdpa_hip.h: a debugging hook for tests/conftest.py and tools/.
SDPA_API int sdpa_debug_dma_audit(unsigned long long out[2]) {
    unsigned long long a[2], b[2];
    sdpa::dma_audit_read_bf16(a);
    sdpa::dma_audit_read_dksplit(b);
    out[0] = a[0] + b[0];
    out[1] = a[1] + b[1];
    return SDPA_OK;
}

const char *sdpa_strerror(int code) {
    switch (code) {
        case SDPA_OK:     return "ok";
        case SDPA_EINVAL: return "invalid argument";
        case SDPA_ENODEV: return "no usable HIP device (this engine has no CPU fallback)";
        case SDPA_EHIP:   return "HIP runtime error";
        case SDPA_ERCCL:  return "collective (RCCL) error";
        case SDPA_ENOMEM: return "out of device or pinned memory";
        case SDPA_EUNSUP: return "shape not supported";
        default:          return "unknown sdpa error";
    }
}

int sdpa_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) {
        (void)hipGetLastError();
        return SDPA_ENODEV;
    }
    return cnt;
}

int sdpa_owner_count(int n, int size, int rank) {
    if (size <= 0) return SDPA_EINVAL;
    const int q = n / size, r = n % size;
    return rank < r ? q + 1 : q;
}

int sdpa_owner_disp(int n, int size, int rank) {
    if (size <= 0) return SDPA_EINVAL;
    const int q = n / size, r = n % size;
    return rank * q + (rank < r ? rank : r);
}

// =============================================================================
// device level
// =============================================================================
int sdpa_dev_cvt_d2f(const double *src, float *dst, long rows, int cols, int ld, void *stream) {
    if (rows < 0 || cols <= 0 || check_ld(ld, cols) != SDPA_OK) return SDPA_EINVAL;
    if (rows == 0) return SDPA_OK;
    if (!src || !dst) return SDPA_EINVAL;
    if (misaligned16(src) || misaligned16(dst)) return SDPA_EINVAL;   // 16-byte accesses on both sides
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_cvt_d2f(src, dst, rows, cols, ld, (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_cvt_d2f_batch(int count, const double *const *src, float *const *dst, const long *rows, const int *cols, const int *ld,
                           void *stream) {
    if (count < 1 || count > 3 || !src || !dst || !rows || !cols || !ld) return SDPA_EINVAL;
    for (int k = 0; k < count; ++k) {
        if (rows[k] < 0 || cols[k] <= 0 || check_ld(ld[k], cols[k])) return SDPA_EINVAL;
        if (rows[k] > 0 && (!src[k] || !dst[k] || misaligned16(src[k]) || misaligned16(dst[k]))) return SDPA_EINVAL;
    }
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_cvt_d2f_batch(count, src, dst, rows, cols, ld, (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_cvt_f2d(const float *src, int ld, double *dst, long rows, int cols, void *stream) {
    if (rows < 0 || cols <= 0 || ld < cols) return SDPA_EINVAL;
    if (rows == 0) return SDPA_OK;
    if (!src || !dst) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_cvt_f2d(src, ld, dst, rows, cols, (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_kv_splits(int m, int n_local, int dk, int dv) {
    if (m <= 0 || n_local < 0 || dk <= 0 || dv <= 0) return SDPA_EINVAL;
    return sdpa::pick_kv_splits(m, n_local, dk, dv);
}

size_t sdpa_dev_workspace_bytes(int m, int n_local, int dk, int dv) {
    if (m <= 0 || n_local < 0 || dk <= 0 || dv <= 0) return 0;
    return sdpa::workspace_bytes(m, n_local, dk, dv);
}

int sdpa_dev_shard_partial_f32(const float *Qf, int ldq, const float *Kf, int ldk, const float *Vf,
                               int ldv, float *contrib, int ldo, float *lmax, float *lsum, int m,
                               int n_local, int dk, int dv, void *workspace, size_t workspace_bytes,
                               void *stream) {
    if (m <= 0 || n_local < 0 || dk <= 0 || dv <= 0) return SDPA_EINVAL;
    if (!Qf || !contrib || !lmax || !lsum) return SDPA_EINVAL;
    if (n_local > 0 && (!Kf || !Vf)) return SDPA_EINVAL;
    if (check_ld(ldq, dk) || check_ld(ldk, dk) || check_ld(ldv, dv) || check_ld(ldo, dv))
        return SDPA_EINVAL;
    if (misaligned16(Qf) || misaligned16(Kf) || misaligned16(Vf) || misaligned16(contrib))
        return SDPA_EINVAL;              // the kernels use 16-byte accesses on every operand
    SDPA_TRY(require_device());
    PartialArgs a = {};
    a.Q = Qf; a.ldq = ldq; a.K = Kf; a.ldk = ldk; a.V = Vf; a.ldv = ldv;
    a.contrib = contrib; a.ldo = ldo; a.lmax = lmax; a.lsum = lsum;
    a.m = m; a.n_local = n_local; a.dk = dk; a.dv = dv;
    // the split count belongs to the STREAM: one with a reservation has fewer workgroup slots to fill, and the
    // stream-K cuts (sdpa_internal.h) follow them.  sdpa_dev_workspace_bytes() covers either.
    a.kv_splits = sdpa::pick_kv_splits(m, n_local, dk, dv, sdpa::stream_cus((hipStream_t)stream));
    if (a.kv_splits > 1 && workspace && workspace_bytes < sdpa::workspace_bytes_for(m, dv, a.kv_splits) &&
        workspace_bytes >= sdpa::workspace_bytes(m, n_local, dk, dv)) {
        // a reserving stream with stream-K switched off wants more equal splits than the documented scratch
        // holds: any smaller count is as correct (only less evenly spread), so take what fits
        while (a.kv_splits > 1 && workspace_bytes < sdpa::workspace_bytes_for(m, dv, a.kv_splits)) --a.kv_splits;
    }
    if (a.kv_splits > 1) {
        if (!workspace || workspace_bytes < sdpa::workspace_bytes_for(m, dv, a.kv_splits)) return SDPA_EINVAL;
        if (misaligned16(workspace)) return SDPA_EINVAL;
        sdpa::carve_workspace(a, workspace, sdpa::dense_ld(dv));
    }
    HIP_TRY(sdpa::launch_shard_partial(a, (hipStream_t)stream));
    return SDPA_OK;
}

// The single-shard call: sdpa_dev_shard_partial_f32 + merge step 5 with gsum = lsum + the fp64 writeback, with the finish FUSED into
// the merge of the in-GPU splits where the launch has splits (one pass over the slabs instead of split_merge, normalise and f2d)
int sdpa_dev_shard_attention_f64(const float *Qf, int ldq, const float *Kf, int ldk, const float *Vf, int ldv, float *contrib,
                                 int ldo, float *lmax, float *lsum, double *result, int m, int n_local, int dk, int dv,
                                 void *workspace, size_t workspace_bytes, void *stream) {
    if (m <= 0 || n_local < 0 || dk <= 0 || dv <= 0) return SDPA_EINVAL;
    if (!Qf || !contrib || !lmax || !lsum || !result) return SDPA_EINVAL;
    if (n_local > 0 && (!Kf || !Vf)) return SDPA_EINVAL;
    if (check_ld(ldq, dk) || check_ld(ldk, dk) || check_ld(ldv, dv) || check_ld(ldo, dv)) return SDPA_EINVAL;
    if (misaligned16(Qf) || misaligned16(Kf) || misaligned16(Vf) || misaligned16(contrib)) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    PartialArgs a = {};
    a.Q = Qf; a.ldq = ldq; a.K = Kf; a.ldk = ldk; a.V = Vf; a.ldv = ldv;
    a.contrib = contrib; a.ldo = ldo; a.lmax = lmax; a.lsum = lsum;
    a.m = m; a.n_local = n_local; a.dk = dk; a.dv = dv;
    a.kv_splits = sdpa::pick_kv_splits(m, n_local, dk, dv, sdpa::stream_cus((hipStream_t)stream));
    if (a.kv_splits > 1 && workspace && workspace_bytes < sdpa::workspace_bytes_for(m, dv, a.kv_splits) &&
        workspace_bytes >= sdpa::workspace_bytes(m, n_local, dk, dv))
        while (a.kv_splits > 1 && workspace_bytes < sdpa::workspace_bytes_for(m, dv, a.kv_splits)) --a.kv_splits;
    if (a.kv_splits > 1) {
        if (!workspace || workspace_bytes < sdpa::workspace_bytes_for(m, dv, a.kv_splits)) return SDPA_EINVAL;
        if (misaligned16(workspace)) return SDPA_EINVAL;
        sdpa::carve_workspace(a, workspace, sdpa::dense_ld(dv));
        a.defer_merge = 1;                                        // the splits stay in their slabs ...
        HIP_TRY(sdpa::launch_shard_partial(a, (hipStream_t)stream));
        a.defer_merge = 0;
        const sdpa::FinishTarget f = {result, nullptr};           // ... and ONE pass merges, normalises and widens them
        HIP_TRY(sdpa::launch_split_merge_finish(a, f, (hipStream_t)stream));
    } else {
        HIP_TRY(sdpa::launch_shard_partial(a, (hipStream_t)stream));
        HIP_TRY(sdpa::launch_finish_f64(contrib, ldo, lsum, result, m, dv, (hipStream_t)stream));
    }
    return SDPA_OK;
}

int sdpa_host_cvt_rows(const double *src, void *dst, long rows, int cols, int ld, int kind, double mult, int flags) {
    if (rows < 0 || cols <= 0 || ld < cols || kind < 0 || kind > 2) return SDPA_EINVAL;
    if (kind == 2 && ld != sdpa::bf16_pad_dk(cols)) return SDPA_EINVAL;       // (the tiled K image: rows of the padded dk)
    if (rows == 0) return SDPA_OK;
    if (!src || !dst) return SDPA_EINVAL;
    sdpa::host_convert_rows(src, dst, rows, cols, ld, kind == 0 ? sdpa::kCvtF32 : kind == 1 ? sdpa::kCvtBf16 : sdpa::kCvtBf16Swz,
                            kind == 0 ? 1.0 : mult, (flags & 1) != 0, (flags & 2) ? 1 : (flags & 4) ? 0 : -1);
    return SDPA_OK;
}

int sdpa_host_cvt_vt(const double *src, unsigned short *dst, long keys, long keys_pad, int cols, int cols_pad, long ldt, int threads,
                     int flags) {
    if (keys < 0 || cols <= 0 || cols_pad < cols || keys_pad < keys || keys_pad % 32 != 0 || ldt < keys_pad) return SDPA_EINVAL;
    if (cols > 256 && cols_pad % 512 != 0) return SDPA_EINVAL;                 // (the tiled image: whole 512-column chunks)
    if (keys_pad == 0) return SDPA_OK;
    if ((!src && keys > 0) || !dst) return SDPA_EINVAL;
    if (threads <= 1) {
        sdpa::host_convert_vt(src, dst, keys, keys_pad, cols, cols_pad, ldt, (flags & 1) != 0, (flags & 2) ? 1 : (flags & 4) ? 0 : -1);
        return SDPA_OK;
    }
    // the way the engine's streamed bf16 call runs it: work items of whole 32-key tiles on a pool of host threads
    sdpa::HostConverter *pool = sdpa::HostConverter::create(threads);
    if (!pool) return SDPA_ENOMEM;
    pool->begin();
    const int task = pool->submit_t(src, dst, keys, keys_pad, cols, cols_pad, ldt);
    pool->kick();
    pool->wait(task);
    pool->finish();
    delete pool;
    return SDPA_OK;
}

int sdpa_host_widen(const float *src, double *dst, size_t n, int threads, int flags) {
    if (n == 0) return SDPA_OK;
    if (!src || !dst) return SDPA_EINVAL;
    if (threads <= 1) {
        sdpa::host_widen(src, dst, n, (flags & 1) != 0);
        return SDPA_OK;
    }
    sdpa::HostConverter *pool = sdpa::HostConverter::create(threads);
    if (!pool) return SDPA_ENOMEM;
    pool->widen(src, dst, n);
    delete pool;
    return SDPA_OK;
}

int sdpa_dev_last_launch(char *buf, size_t len) {
    if (!buf || len == 0) return SDPA_EINVAL;
    const sdpa::LaunchNote &n = sdpa::last_launch_note();
    char name[128];
    sdpa::format_launch_kernel(n, name, sizeof name);
    const int need = snprintf(buf, len, "{\"kernel\": \"%s\", \"grid\": %d, \"splits\": %d, \"stream_k\": %d, \"rows\": %d, "
                              "\"keys\": %d}", name, n.grid, n.splits, n.streamk, n.rows, n.keys);
    return (need < 0 || (size_t)need >= len) ? SDPA_EINVAL : SDPA_OK;
}

int sdpa_dev_stream_create(int reserve_cus, void **stream) {
    if (!stream || reserve_cus < 0) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    hipStream_t st = nullptr;
    SDPA_TRY(sdpa::create_masked_stream(&st, reserve_cus));
    *stream = (void *)st;
    return SDPA_OK;
}

int sdpa_dev_stream_destroy(void *stream) {
    if (!stream) return SDPA_EINVAL;
    sdpa::forget_stream_cus((hipStream_t)stream);
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_dense_ld(int d) { return d < 1 ? 0 : sdpa::dense_ld(d); }

int sdpa_dev_merge_rescale(float *contrib, int ldo, float *lsum, const float *lmax, const float *gmax,
                           int m, int dv, void *stream) {
    if (!contrib || !lsum || !lmax || !gmax || m <= 0 || dv <= 0 || check_ld(ldo, dv)) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_merge_rescale(contrib, ldo, lsum, lmax, gmax, m, dv, (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_merge_normalise(float *contrib, int ldo, const float *gsum, int m, int dv, void *stream) {
    if (!contrib || !gsum || m <= 0 || dv <= 0 || check_ld(ldo, dv)) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_merge_normalise(contrib, ldo, gsum, m, dv, (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_merge_gathered(float *contrib, int ldo, const float *stats, int parts, int self, int m,
                            int dv, void *stream) {
    if (!contrib || !stats || parts <= 0 || self < 0 || self >= parts || m <= 0 || dv <= 0 ||
        check_ld(ldo, dv))
        return SDPA_EINVAL;
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_merge_gathered(contrib, ldo, stats, parts, self, m, dv, (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_finish_f64(const float *contrib, int ldo, const float *lsum, double *result, int m,
                        int dv, void *stream) {
    if (!contrib || !lsum || !result || m <= 0 || dv <= 0 || check_ld(ldo, dv)) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_finish_f64(contrib, ldo, lsum, result, m, dv, (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_bf16_ld(int dk) { return (dk <= 0 || dk > 512) ? SDPA_EUNSUP : sdpa::bf16_pad_dk(dk); }
int sdpa_dev_bf16_dvp(int dv) { return (dv <= 0 || dv > 1024) ? SDPA_EUNSUP : sdpa::bf16_pad_dv(dv); }
long sdpa_dev_bf16_ldn(long n_local) { return n_local < 0 ? SDPA_EINVAL : sdpa::bf16_pad_n(n_local); }
long sdpa_dev_bf16_kvpos(long j) { return j < 0 ? SDPA_EINVAL : sdpa::bf16_kvpos(j); }

int sdpa_dev_cvt_d2bf(const double *src, void *dst, long rows, int cols, int ld, void *stream) {
    if (rows < 0 || cols <= 0 || ld < cols) return SDPA_EINVAL;
    if (rows == 0) return SDPA_OK;
    if (!src || !dst) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_cvt_d2bf(src, (unsigned short *)dst, rows, cols, ld, (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_bf16_tiled(int dv) { return dv <= 0 ? SDPA_EINVAL : (sdpa::bf16_tiled(dv) ? 1 : 0); }

int sdpa_dev_cvt_d2bf_k(const double *src, void *dst, long rows, int dk, int dv, void *stream) {
    if (rows < 0 || dk <= 0 || dk > 512 || dv <= 0 || dv > 1024) return SDPA_EINVAL;
    if (rows == 0) return SDPA_OK;
    if (!src || !dst) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_cvt_d2bf_k(src, (unsigned short *)dst, rows, sdpa::bf16_tiled(dv) ? sdpa::bf16_pad_n(rows) : rows, dk, dv,
                                    (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_cvt_d2bf_v(const double *src, void *dst, long rows, int dv, void *stream) {
    if (rows < 0 || dv <= 0 || dv > 1024) return SDPA_EINVAL;
    if (!dst || (rows > 0 && !src)) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_cvt_d2bf_t(src, (unsigned short *)dst, rows, dv, sdpa::bf16_pad_dv(dv), sdpa::bf16_pad_n(rows),
                                    (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_cvt_d2bf_q(const double *src, void *dst, long rows, int dk, int ld, void *stream) {
    if (rows < 0 || dk <= 0 || dk > 512 || ld != sdpa::bf16_pad_dk(dk)) return SDPA_EINVAL;
    if (rows == 0) return SDPA_OK;
    if (!src || !dst) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_cvt_d2bf_q(src, (unsigned short *)dst, rows, dk, ld, (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_cvt_d2bf_t(const double *src, void *dst, long rows, int cols, int cols_pad, long ldt,
                        void *stream) {
    if (rows < 0 || cols <= 0 || cols_pad < cols || ldt < rows) return SDPA_EINVAL;
    // keys are permuted inside 16-key groups and written in 32-key blocks: a row stride that is
    // not a multiple of 32 would let the last block spill into the next Vt row
    if (ldt % 32 != 0) return SDPA_EINVAL;
    if (sdpa::bf16_tiled(cols) && cols_pad % 512 != 0) return SDPA_EINVAL;    // (the tiled image: whole 512-column chunks)
    if (!dst || (rows > 0 && !src)) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_cvt_d2bf_t(src, (unsigned short *)dst, rows, cols, cols_pad, ldt,
                                    (hipStream_t)stream));
    return SDPA_OK;
}

int sdpa_dev_kv_splits_bf16(int m, int n_local, int dk, int dv) {
    if (m <= 0 || n_local < 0 || dk <= 0 || dv <= 0) return SDPA_EINVAL;
    return sdpa::pick_kv_splits_bf16(m, n_local, dk, dv);
}

size_t sdpa_dev_workspace_bytes_bf16(int m, int n_local, int dk, int dv) {
    if (m <= 0 || n_local < 0 || dk <= 0 || dv <= 0) return 0;
    return sdpa::bf16_workspace_bytes(m, n_local, dk, dv);
}

int sdpa_dev_shard_partial_bf16(const void *Qb, int ldq, const void *Kb, int ldk, const void *Vt,
                                long ldvt, float *contrib, int ldo, float *lmax, float *lsum, int m,
                                int n_local, int dk, int dv, void *workspace, size_t workspace_bytes,
                                void *stream) {
    if (m <= 0 || n_local <= 0 || dk <= 0 || dv <= 0) return SDPA_EINVAL;
    if (!Qb || !Kb || !Vt || !contrib || !lmax || !lsum) return SDPA_EINVAL;
    if (dk > 512 || dv > 1024) return SDPA_EUNSUP;
    if (ldq != sdpa::bf16_pad_dk(dk) || ldk != ldq || ldvt < n_local || ldvt % 32 != 0 || check_ld(ldo, dv))
        return SDPA_EINVAL;
    if (misaligned16(Qb) || misaligned16(Kb) || misaligned16(Vt) || misaligned16(contrib)) return SDPA_EINVAL;
    if ((double)sdpa::bf16_pad_dv(dv) * (double)ldvt * 2.0 >= 4294967296.0) return SDPA_EUNSUP;
    SDPA_TRY(require_device());
    sdpa::Bf16Args a = {};
    a.Q = (const unsigned short *)Qb; a.ldq = ldq; a.K = (const unsigned short *)Kb; a.ldk = ldk;
    a.Vt = (const unsigned short *)Vt; a.ldvt = ldvt;
    a.contrib = contrib; a.ldo = ldo; a.lmax = lmax; a.lsum = lsum;
    a.m = m; a.n_local = n_local; a.dk = dk; a.dv = dv;
    a.kv_splits = sdpa::pick_kv_splits_bf16(m, n_local, dk, dv);
    const size_t need = sdpa::bf16_workspace_bytes(m, n_local, dk, dv);
    if (need && (!workspace || workspace_bytes < need)) return SDPA_EINVAL;
    sdpa::bf16_carve_workspace(a, workspace, round4(dv));
    HIP_TRY(sdpa::launch_shard_partial_bf16(a, (hipStream_t)stream));
    return SDPA_OK;
}

}  // extern "C"
