// sdpa_api.hip -- the C ABI declared in include/sdpa_hip.h.
//
// Host level: sdpa_attention_f64() is the body of the reference's attention()
// (attention.c:20-75 / attention-mpi.c:191-407) re-done for one process driving
// 1..8 MI355X:
//   attention-mpi.c:210-266  K/V convert + Bcast/Scatterv  -> per-GPU shard H2D + device convert
//   attention-mpi.c:268-330  Q ping-pong + MPI_Ibcast      -> two Q slots, copy stream ahead of
//                                                            the compute stream, ncclBroadcast
//   attention-mpi.c:333-338  per-row online softmax         -> the fused kernel (sdpa_fwd_f32.hip)
//   attention-mpi.c:340-362  Iallreduce MAX / SUM + scales  -> ncclAllReduce(ncclMax|ncclSum) + merge kernels
//   attention-mpi.c:364-399  Ireduce + f2d writeback        -> ncclReduce + convert + D2H on the out stream
// Device level: thin argument-checking wrappers over the launchers.
//
// RCCL is bound with dlopen at sdpa_init(n > 1) only: a single-GPU engine never
// loads it, and inside a PyTorch process the loader hands back the librccl.so.1
// PyTorch already mapped instead of a second copy.
#include "../../include/sdpa_hip.h"
#include "sdpa_internal.h"

#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

namespace {

using sdpa::PartialArgs;

// ---- minimal RCCL surface (NCCL API), resolved at run time ------------------
typedef struct ncclComm *ncclComm_t;
enum { kNcclSuccess = 0 };
enum { kNcclUint8 = 1, kNcclFloat = 7 };  // ncclUint8, ncclFloat32
enum { kNcclSum = 0, kNcclMax = 2 };     // ncclRedOp_t
struct Rccl {
    void *handle = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

template <class F>
bool bind(void *h, const char *name, F &fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

bool load_rccl(Rccl &r) {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) {
        fprintf(stderr, "sdpa: cannot load RCCL: %s\n", dlerror());
        return false;
    }
    return bind(r.handle, "ncclCommInitAll", r.CommInitAll) &&
           bind(r.handle, "ncclCommDestroy", r.CommDestroy) &&
           bind(r.handle, "ncclGroupStart", r.GroupStart) &&
           bind(r.handle, "ncclGroupEnd", r.GroupEnd) &&
           bind(r.handle, "ncclAllReduce", r.AllReduce) &&
           bind(r.handle, "ncclReduce", r.Reduce) &&
           bind(r.handle, "ncclBroadcast", r.Broadcast) &&
           bind(r.handle, "ncclGetErrorString", r.GetErrorString);
}

// ---- error plumbing ----------------------------------------------------------
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "sdpa: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                   \
            return e_ == hipErrorOutOfMemory ? SDPA_ENOMEM : SDPA_EHIP;                    \
        }                                                                                  \
    } while (0)

#define RCCL_TRY(expr)                                                                  \
    do {                                                                                \
        int r_ = (expr);                                                                \
        if (r_ != kNcclSuccess) {                                                       \
            fprintf(stderr, "sdpa: %s failed: %s (%s:%d)\n", #expr,                     \
                    E.rccl.GetErrorString ? E.rccl.GetErrorString(r_) : "?", __FILE__,  \
                    __LINE__);                                                          \
            return SDPA_ERCCL;                                                          \
        }                                                                               \
    } while (0)

#define SDPA_TRY(expr)          \
    do {                        \
        int c_ = (expr);        \
        if (c_ != SDPA_OK) return c_; \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

int ensure(DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return SDPA_OK;
    if (b.p) HIP_TRY(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    HIP_TRY(hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return SDPA_OK;
}

struct Gpu {
    int dev = 0;
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
    ncclComm_t comm = nullptr;
    DevBuf k64, v64, kf, vf, ws;
    DevBuf q64[2], qf[2], contrib[2], lmax[2], lsum[2], gmax[2], gsum[2], red[2], out64[2];
    hipEvent_t ev_q[2] = {}, ev_run[2] = {}, ev_out[2] = {};
    std::vector<hipEvent_t> ev_k;   // fused-kernel timing brackets, 2 per Q batch (GPU 0)
};

struct Engine {
    bool up = false;
    int n = 0;
    std::vector<Gpu> g;
    Rccl rccl;
    sdpa_timing last = {};
};
// Heap-allocated and never destroyed on purpose: at process exit the order in which this library's
// static destructors and the HIP runtime's run is not ours to choose (inside a Python process the
// runtime belongs to PyTorch), and nothing here needs tearing down then -- sdpa_shutdown() is the
// explicit release.
Engine &E = *new Engine;

inline int round4(int x) { return (x + 3) / 4 * 4; }

// RAII page-locking of caller-owned host arrays (best effort: a range that cannot be registered,
// e.g. because the caller already did, is simply left as it is).
struct HostPins {
    void *ptr[4];
    int n = 0;
    void add(const void *p, size_t bytes) {
        if (bytes < (1u << 20) || n >= 4) return;          // small arrays: not worth the call
        if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) == hipSuccess)
            ptr[n++] = const_cast<void *>(p);
        else
            (void)hipGetLastError();
    }
    ~HostPins() {
        for (int i = 0; i < n; ++i)
            if (hipHostUnregister(ptr[i]) != hipSuccess) (void)hipGetLastError();
    }
};

double now_us() {
    using namespace std::chrono;
    return duration<double, std::micro>(steady_clock::now().time_since_epoch()).count();
}

int check_ld(int ld, int cols) { return (ld >= cols && ld % 4 == 0) ? SDPA_OK : SDPA_EINVAL; }

bool misaligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

int require_device() {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
        (void)hipGetLastError();
        return SDPA_ENODEV;
    }
    return SDPA_OK;
}

}  // namespace

// =============================================================================
// lifecycle
// =============================================================================
extern "C" {

const char *sdpa_version(void) { return "sdpa-hip 0.1 (gfx950, f32 MFMA)"; }

const char *sdpa_strerror(int code) {
    switch (code) {
        case SDPA_OK:     return "ok";
        case SDPA_EINVAL: return "invalid argument";
        case SDPA_ENODEV: return "no usable HIP device (this engine has no CPU fallback)";
        case SDPA_EHIP:   return "HIP runtime error";
        case SDPA_ERCCL:  return "RCCL error";
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

void sdpa_shutdown(void) {
    if (!E.up) return;
    for (Gpu &g : E.g) {
        if (hipSetDevice(g.dev) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
        if (g.comm && E.rccl.CommDestroy) E.rccl.CommDestroy(g.comm);
        DevBuf *single[] = {&g.k64, &g.v64, &g.kf, &g.vf, &g.ws};
        for (DevBuf *b : single) if (b->p) (void)hipFree(b->p);
        for (int s = 0; s < 2; ++s) {
            DevBuf *pair[] = {&g.q64[s], &g.qf[s], &g.contrib[s], &g.lmax[s], &g.lsum[s],
                              &g.gmax[s], &g.gsum[s], &g.red[s], &g.out64[s]};
            for (DevBuf *b : pair) if (b->p) (void)hipFree(b->p);
            if (g.ev_q[s]) (void)hipEventDestroy(g.ev_q[s]);
            if (g.ev_run[s]) (void)hipEventDestroy(g.ev_run[s]);
            if (g.ev_out[s]) (void)hipEventDestroy(g.ev_out[s]);
        }
        for (hipEvent_t e : g.ev_k) (void)hipEventDestroy(e);
        if (g.s_in) (void)hipStreamDestroy(g.s_in);
        if (g.s_run) (void)hipStreamDestroy(g.s_run);
        if (g.s_out) (void)hipStreamDestroy(g.s_out);
    }
    E.g.clear();
    E.n = 0;
    E.up = false;
}

int sdpa_init(int n_gpus) {
    if (n_gpus < 0) return SDPA_EINVAL;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
        (void)hipGetLastError();
        fprintf(stderr, "sdpa: no HIP device visible\n");
        return SDPA_ENODEV;
    }
    const int want = n_gpus == 0 ? cnt : n_gpus;
    if (want > cnt) {
        fprintf(stderr, "sdpa: %d GPUs requested, %d visible\n", want, cnt);
        return SDPA_ENODEV;
    }
    if (E.up && E.n == want) return SDPA_OK;
    if (E.up) sdpa_shutdown();

    E.g.assign(want, Gpu());
    for (int i = 0; i < want; ++i) {
        Gpu &g = E.g[i];
        g.dev = i;
        HIP_TRY(hipSetDevice(i));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, i));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            fprintf(stderr, "sdpa: device %d is %s; this engine is built for gfx950 only\n", i,
                    prop.gcnArchName);
            return SDPA_ENODEV;
        }
        HIP_TRY(hipStreamCreateWithFlags(&g.s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&g.s_run, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&g.s_out, hipStreamNonBlocking));
        for (int s = 0; s < 2; ++s) {
            HIP_TRY(hipEventCreateWithFlags(&g.ev_q[s], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&g.ev_run[s], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&g.ev_out[s], hipEventDisableTiming));
        }
    }
    if (want > 1) {
        if (!E.rccl.handle && !load_rccl(E.rccl)) return SDPA_ERCCL;
        std::vector<ncclComm_t> comms(want);
        std::vector<int> devs(want);
        for (int i = 0; i < want; ++i) devs[i] = i;
        RCCL_TRY(E.rccl.CommInitAll(comms.data(), want, devs.data()));
        for (int i = 0; i < want; ++i) E.g[i].comm = comms[i];
    }
    HIP_TRY(hipSetDevice(0));
    E.n = want;
    E.up = true;
    return SDPA_OK;
}

int sdpa_last_timing(struct sdpa_timing *out) {
    if (!out) return SDPA_EINVAL;
    *out = E.last;
    return SDPA_OK;
}

// =============================================================================
// host level
// =============================================================================
// Device buffers of the Q-batch pipeline for this shape (grow-only, cached across calls).
static int ensure_batch_buffers(int m, int n, int dk, int dv, bool bf16, int B, int nb) {
    const int P = E.n;
    const int ldo = round4(dv);
    const int ldq = bf16 ? sdpa::bf16_pad_dk(dk) : round4(dk);
    const size_t q_elem = bf16 ? sizeof(unsigned short) : sizeof(float);
    for (int i = 0; i < P; ++i) {
        Gpu &g = E.g[i];
        HIP_TRY(hipSetDevice(g.dev));
        const int cnt = sdpa_owner_count(n, P, i);
        const int tail_rows = m - (nb - 1) * B;
        const size_t ws_full = bf16 ? sdpa_dev_workspace_bytes_bf16(B, cnt, dk, dv) : sdpa::workspace_bytes(B, cnt, dk, dv);
        const size_t ws_tail = bf16 ? sdpa_dev_workspace_bytes_bf16(tail_rows, cnt, dk, dv)
                                    : sdpa::workspace_bytes(tail_rows, cnt, dk, dv);
        SDPA_TRY(ensure(g.ws, ws_full > ws_tail ? ws_full : ws_tail));
        for (int s = 0; s < 2; ++s) {
            SDPA_TRY(ensure(g.qf[s], (size_t)B * ldq * q_elem));
            SDPA_TRY(ensure(g.contrib[s], (size_t)B * ldo * sizeof(float)));
            SDPA_TRY(ensure(g.lmax[s], (size_t)B * sizeof(float)));
            SDPA_TRY(ensure(g.lsum[s], (size_t)B * sizeof(float)));
            if (P > 1) {
                SDPA_TRY(ensure(g.gmax[s], (size_t)B * sizeof(float)));
                SDPA_TRY(ensure(g.gsum[s], (size_t)B * sizeof(float)));
            }
            if (i == 0) {
                SDPA_TRY(ensure(g.q64[s], (size_t)B * dk * sizeof(double)));
                SDPA_TRY(ensure(g.out64[s], (size_t)B * dv * sizeof(double)));
                if (P > 1) SDPA_TRY(ensure(g.red[s], (size_t)B * ldo * sizeof(float)));
            }
        }
    }
    return SDPA_OK;
}

// Staging buffers of one GPU's K/V shard (grow-only).
static int ensure_kv_buffers(Gpu &g, int n, int dk, int dv, int P, bool bf16) {
    const int cnt = sdpa_owner_count(n, P, g.dev);
    HIP_TRY(hipSetDevice(g.dev));
    SDPA_TRY(ensure(g.k64, (size_t)cnt * dk * sizeof(double)));
    SDPA_TRY(ensure(g.v64, (size_t)cnt * dv * sizeof(double)));
    if (bf16) {
        SDPA_TRY(ensure(g.kf, (size_t)cnt * sdpa::bf16_pad_dk(dk) * sizeof(unsigned short)));
        SDPA_TRY(ensure(g.vf, (size_t)sdpa::bf16_pad_dv(dv) * sdpa::bf16_pad_n(cnt) * sizeof(unsigned short)));
    } else {
        SDPA_TRY(ensure(g.kf, (size_t)cnt * round4(dk) * sizeof(float)));
        SDPA_TRY(ensure(g.vf, (size_t)cnt * round4(dv) * sizeof(float)));
    }
    return SDPA_OK;
}

static int pick_q_batch(int m, int flags) {
    int B = 8192;
    if (const char *env = getenv("SDPA_QBATCH")) B = atoi(env) > 0 ? atoi(env) : B;
    if ((flags & SDPA_F_NO_PIPELINE) || B > m) B = m;
    return B;
}

static bool want_bf16(int flags) {
    bool bf16 = (flags & SDPA_F_BF16) != 0;
    if (const char *prec = getenv("SDPA_PRECISION")) bf16 = bf16 || strcmp(prec, "bf16") == 0;
    return bf16;
}

// K/V rows [owner_disp, +owner_count) of this GPU: host -> device, convert to the operand image
// (attention-mpi.c:224-225 / :248-249 and the Scatterv of :258-264).
static int stage_kv_shard(Gpu &g, const double *K, const double *V, int n, int dk, int dv, int P,
                          bool bf16) {
    const int cnt = sdpa_owner_count(n, P, g.dev);
    const int off = sdpa_owner_disp(n, P, g.dev);
    const int ldk = round4(dk), ldv = round4(dv);
    SDPA_TRY(ensure_kv_buffers(g, n, dk, dv, P, bf16));
    const int ldb = sdpa::bf16_pad_dk(dk), dvp = sdpa::bf16_pad_dv(dv);
    const long ldn = sdpa::bf16_pad_n(cnt);
    if (cnt > 0) {
        HIP_TRY(hipMemcpyAsync(g.k64.p, K + (size_t)off * dk, (size_t)cnt * dk * sizeof(double),
                               hipMemcpyHostToDevice, g.s_in));
        if (bf16)
            HIP_TRY(sdpa::launch_cvt_d2bf((const double *)g.k64.p, (unsigned short *)g.kf.p, cnt, dk, ldb, g.s_in));
        else
            HIP_TRY(sdpa::launch_cvt_d2f((const double *)g.k64.p, (float *)g.kf.p, cnt, dk, ldk, g.s_in));
        HIP_TRY(hipMemcpyAsync(g.v64.p, V + (size_t)off * dv, (size_t)cnt * dv * sizeof(double),
                               hipMemcpyHostToDevice, g.s_in));
        if (bf16)
            HIP_TRY(sdpa::launch_cvt_d2bf_t((const double *)g.v64.p, (unsigned short *)g.vf.p, cnt, dv, dvp, ldn, g.s_in));
        else
            HIP_TRY(sdpa::launch_cvt_d2f((const double *)g.v64.p, (float *)g.vf.p, cnt, dv, ldv, g.s_in));
    }
    HIP_TRY(hipStreamSynchronize(g.s_in));
    return SDPA_OK;
}

int sdpa_attention_f64(const double *Q, const double *K, const double *V, double *result, int m,
                       int n, int dk, int dv, int flags) {
    if (!Q || !K || !V || !result || m <= 0 || n <= 0 || dk <= 0 || dv <= 0) return SDPA_EINVAL;
    if (dv > 1024) return SDPA_EUNSUP;
    const bool bf16 = want_bf16(flags);
    if (bf16 && dk > 512) return SDPA_EUNSUP;
    const double t_enter = now_us();
    if (!E.up) {
        const char *env = getenv("SDPA_GPUS");
        SDPA_TRY(sdpa_init(env ? atoi(env) : 0));
    }
    // Page-lock the caller's arrays for the duration of the call.  Copies from pages the driver
    // has never seen run at ~11 GB/s on this platform (measured, tools/probes/h2d_probe.cpp);
    // registering 134 MB costs ~2 ms and the copies then run at ~57 GB/s and truly
    // asynchronously.  Nothing stays registered after the call (no pointer is retained).
    HostPins pins;
    {
        const char *env = getenv("SDPA_HOST_REGISTER");
        if (!env || atoi(env) != 0) {
            pins.add(Q, (size_t)m * dk * sizeof(double));
            pins.add(K, (size_t)n * dk * sizeof(double));
            pins.add(V, (size_t)n * dv * sizeof(double));
            pins.add(result, (size_t)m * dv * sizeof(double));
        }
    }
    const int P = E.n;
    const int ldo = round4(dv);
    const int ldq = bf16 ? sdpa::bf16_pad_dk(dk) : round4(dk);      // elements per staged Q row
    const size_t q_elem = bf16 ? sizeof(unsigned short) : sizeof(float);

    // ---- K/V shards: rows [owner_disp, +owner_count) of K and V to GPU g ---------------
    // One host thread per GPU so the PCIe links work in parallel even from pageable memory.
    {
        std::vector<int> rc(P, SDPA_OK);
        if (P == 1) {
            rc[0] = stage_kv_shard(E.g[0], K, V, n, dk, dv, P, bf16);
        } else {
            std::vector<std::thread> th;
            for (int i = 0; i < P; ++i)
                th.emplace_back([&, i] { rc[i] = stage_kv_shard(E.g[i], K, V, n, dk, dv, P, bf16); });
            for (auto &t : th) t.join();
        }
        for (int i = 0; i < P; ++i) SDPA_TRY(rc[i]);
    }
    const double t_kv = now_us();

    // ---- Q batches -------------------------------------------------------------------
    const int B = pick_q_batch(m, flags);
    const int nb = (m + B - 1) / B;
    int splits_used = 1;
    double kernel_ms = 0.0;

    SDPA_TRY(ensure_batch_buffers(m, n, dk, dv, bf16, B, nb));

    Gpu &root = E.g[0];
    HIP_TRY(hipSetDevice(root.dev));
    while ((int)root.ev_k.size() < 2 * nb) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        root.ev_k.push_back(e);
    }
    auto drain_batch = [&](int b) -> int {   // result rows of batch b: device -> caller
        const int s = b & 1, i0 = b * B, bs = (i0 + B <= m) ? B : m - i0;
        HIP_TRY(hipSetDevice(root.dev));
        HIP_TRY(hipStreamWaitEvent(root.s_out, root.ev_run[s], 0));
        HIP_TRY(hipMemcpyAsync(result + (size_t)i0 * dv, root.out64[s].p,
                               (size_t)bs * dv * sizeof(double), hipMemcpyDeviceToHost, root.s_out));
        HIP_TRY(hipEventRecord(root.ev_out[s], root.s_out));
        return SDPA_OK;
    };

    for (int b = 0; b < nb; ++b) {
        const int s = b & 1, i0 = b * B, bs = (i0 + B <= m) ? B : m - i0;

        // copy stream (root): Q batch fp64 -> device, convert.  Slot s was last read by
        // the compute of batch b-2 (ev_run[s]).
        HIP_TRY(hipSetDevice(root.dev));
        if (b >= 2) HIP_TRY(hipStreamWaitEvent(root.s_in, root.ev_run[s], 0));
        HIP_TRY(hipMemcpyAsync(root.q64[s].p, Q + (size_t)i0 * dk, (size_t)bs * dk * sizeof(double),
                               hipMemcpyHostToDevice, root.s_in));
        if (bf16)
            HIP_TRY(sdpa::launch_cvt_d2bf((const double *)root.q64[s].p, (unsigned short *)root.qf[s].p, bs,
                                          dk, ldq, root.s_in));
        else
            HIP_TRY(sdpa::launch_cvt_d2f((const double *)root.q64[s].p, (float *)root.qf[s].p, bs, dk,
                                         ldq, root.s_in));
        HIP_TRY(hipEventRecord(root.ev_q[s], root.s_in));
        HIP_TRY(hipStreamWaitEvent(root.s_run, root.ev_q[s], 0));
        // out64[s] is still being copied out for batch b-2
        if (b >= 2) HIP_TRY(hipStreamWaitEvent(root.s_run, root.ev_out[s], 0));

        if (P > 1) {   // MPI_Ibcast of the Q batch (attention-mpi.c:305,:327)
            RCCL_TRY(E.rccl.GroupStart());
            for (int i = 0; i < P; ++i)
                RCCL_TRY(E.rccl.Broadcast(root.qf[s].p, E.g[i].qf[s].p, (size_t)bs * ldq * q_elem, kNcclUint8,
                                          0, E.g[i].comm, E.g[i].s_run));
            RCCL_TRY(E.rccl.GroupEnd());
        }

        // fused kernel on every GPU's shard
        for (int i = 0; i < P; ++i) {
            Gpu &g = E.g[i];
            HIP_TRY(hipSetDevice(g.dev));
            const int n_loc = sdpa_owner_count(n, P, i);
            int splits_here = 1;
            if (i == 0) HIP_TRY(hipEventRecord(g.ev_k[2 * b], g.s_run));
            if (bf16) {
                sdpa::Bf16Args a = {};
                a.Q = (const unsigned short *)g.qf[s].p;  a.ldq = ldq;
                a.K = (const unsigned short *)g.kf.p;     a.ldk = ldq;
                a.Vt = (const unsigned short *)g.vf.p;    a.ldvt = sdpa::bf16_pad_n(n_loc);
                a.contrib = (float *)g.contrib[s].p;  a.ldo = ldo;
                a.lmax = (float *)g.lmax[s].p;
                a.lsum = (float *)g.lsum[s].p;
                a.m = bs;  a.n_local = n_loc;  a.dk = dk;  a.dv = dv;
                a.kv_splits = splits_here = sdpa::pick_kv_splits_bf16(bs, n_loc, dk, dv);
                sdpa::bf16_carve_workspace(a, g.ws.p, ldo);
                if (n_loc > 0) {
                    HIP_TRY(sdpa::launch_shard_partial_bf16(a, g.s_run));
                } else {   // empty shard: the fp32 launcher's T = 0 path writes (0, -inf, 0)
                    PartialArgs e = {};
                    e.Q = (const float *)g.qf[s].p;  e.ldq = 4;  e.ldk = 4;  e.ldv = 4;
                    e.contrib = a.contrib;  e.ldo = ldo;  e.lmax = a.lmax;  e.lsum = a.lsum;
                    e.m = bs;  e.n_local = 0;  e.dk = 4;  e.dv = dv;  e.kv_splits = 1;
                    HIP_TRY(sdpa::launch_shard_partial(e, g.s_run));
                }
            } else {
                PartialArgs a = {};
                a.Q = (const float *)g.qf[s].p;  a.ldq = ldq;
                a.K = (const float *)g.kf.p;     a.ldk = round4(dk);
                a.V = (const float *)g.vf.p;     a.ldv = round4(dv);
                a.contrib = (float *)g.contrib[s].p;  a.ldo = ldo;
                a.lmax = (float *)g.lmax[s].p;
                a.lsum = (float *)g.lsum[s].p;
                a.m = bs;  a.n_local = n_loc;  a.dk = dk;  a.dv = dv;
                a.kv_splits = splits_here = sdpa::pick_kv_splits(bs, a.n_local, dk, dv);
                if (a.kv_splits > 1) {
                    a.ws_ld = ldo;
                    a.ws_contrib = (float *)g.ws.p;
                    a.ws_lmax = a.ws_contrib + (size_t)a.kv_splits * bs * a.ws_ld;
                    a.ws_lsum = a.ws_lmax + (size_t)a.kv_splits * bs;
                }
                HIP_TRY(sdpa::launch_shard_partial(a, g.s_run));
            }
            if (i == 0) splits_used = splits_here;
            if (i == 0) HIP_TRY(hipEventRecord(g.ev_k[2 * b + 1], g.s_run));
        }

        if (P == 1) {
            HIP_TRY(sdpa::launch_finish_f64((const float *)root.contrib[s].p, ldo,
                                            (const float *)root.lsum[s].p, (double *)root.out64[s].p,
                                            bs, dv, root.s_run));
        } else {
            // two-phase merge, attention-mpi.c:340-362, then the reduce of :380
            RCCL_TRY(E.rccl.GroupStart());
            for (int i = 0; i < P; ++i)
                RCCL_TRY(E.rccl.AllReduce(E.g[i].lmax[s].p, E.g[i].gmax[s].p, bs, kNcclFloat, kNcclMax,
                                          E.g[i].comm, E.g[i].s_run));
            RCCL_TRY(E.rccl.GroupEnd());
            for (int i = 0; i < P; ++i) {
                Gpu &g = E.g[i];
                HIP_TRY(hipSetDevice(g.dev));
                HIP_TRY(sdpa::launch_merge_rescale((float *)g.contrib[s].p, ldo, (float *)g.lsum[s].p,
                                                   (const float *)g.lmax[s].p,
                                                   (const float *)g.gmax[s].p, bs, dv, g.s_run));
            }
            RCCL_TRY(E.rccl.GroupStart());
            for (int i = 0; i < P; ++i)
                RCCL_TRY(E.rccl.AllReduce(E.g[i].lsum[s].p, E.g[i].gsum[s].p, bs, kNcclFloat, kNcclSum,
                                          E.g[i].comm, E.g[i].s_run));
            RCCL_TRY(E.rccl.GroupEnd());
            for (int i = 0; i < P; ++i) {
                Gpu &g = E.g[i];
                HIP_TRY(hipSetDevice(g.dev));
                HIP_TRY(sdpa::launch_merge_normalise((float *)g.contrib[s].p, ldo,
                                                     (const float *)g.gsum[s].p, bs, dv, g.s_run));
            }
            RCCL_TRY(E.rccl.GroupStart());
            for (int i = 0; i < P; ++i)
                RCCL_TRY(E.rccl.Reduce(E.g[i].contrib[s].p, i == 0 ? root.red[s].p : nullptr,
                                       (size_t)bs * ldo, kNcclFloat, kNcclSum, 0, E.g[i].comm,
                                       E.g[i].s_run));
            RCCL_TRY(E.rccl.GroupEnd());
            HIP_TRY(hipSetDevice(root.dev));
            HIP_TRY(sdpa::launch_cvt_f2d((const float *)root.red[s].p, ldo, (double *)root.out64[s].p,
                                         bs, dv, root.s_run));
        }
        HIP_TRY(hipSetDevice(root.dev));
        HIP_TRY(hipEventRecord(root.ev_run[s], root.s_run));

        // The D2H into the caller's pageable array blocks this thread, so issue it one
        // batch late: the next batch's work is already queued behind it on the GPU.
        if (b >= 1) SDPA_TRY(drain_batch(b - 1));
    }
    SDPA_TRY(drain_batch(nb - 1));

    for (int i = 0; i < P; ++i) {
        HIP_TRY(hipSetDevice(E.g[i].dev));
        HIP_TRY(hipStreamSynchronize(E.g[i].s_run));
    }
    HIP_TRY(hipSetDevice(root.dev));
    HIP_TRY(hipStreamSynchronize(root.s_out));
    for (int b = 0; b < nb; ++b) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, root.ev_k[2 * b], root.ev_k[2 * b + 1]));
        kernel_ms += ms;
    }
    const double t_exit = now_us();
    E.last.total_us = t_exit - t_enter;
    E.last.kv_stage_us = t_kv - t_enter;
    E.last.pipeline_us = t_exit - t_kv;
    E.last.kernel_us = kernel_ms * 1e3;
    E.last.n_gpus = P;
    E.last.q_batches = nb;
    E.last.kv_splits = splits_used;
    return SDPA_OK;
}

void *sdpa_host_alloc(size_t bytes) {
    if (bytes == 0 || require_device() != SDPA_OK) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void sdpa_host_free(void *p) {
    if (p && hipHostFree(p) != hipSuccess) (void)hipGetLastError();
}

int sdpa_prepare(int m, int n, int dk, int dv, int flags) {
    if (m <= 0 || n <= 0 || dk <= 0 || dv <= 0) return SDPA_EINVAL;
    if (dv > 1024) return SDPA_EUNSUP;
    const bool bf16 = want_bf16(flags);
    if (bf16 && dk > 512) return SDPA_EUNSUP;
    if (!E.up) {
        const char *env = getenv("SDPA_GPUS");
        SDPA_TRY(sdpa_init(env ? atoi(env) : 0));
    }
    // 1. every device buffer the real call will use, at its real size
    for (int i = 0; i < E.n; ++i) SDPA_TRY(ensure_kv_buffers(E.g[i], n, dk, dv, E.n, bf16));
    const int B = pick_q_batch(m, flags);
    SDPA_TRY(ensure_batch_buffers(m, n, dk, dv, bf16, B, (m + B - 1) / B));
    // 2. one small call through the same code path: loads the code objects, sets the kernel
    //    attributes, creates the timing events (the kernel variants depend on dk, dv only)
    const int m0 = m < 256 ? m : 256, n0 = n < 2048 ? n : 2048;
    std::vector<double> q((size_t)m0 * dk, 0.25), k((size_t)n0 * dk, 0.5), v((size_t)n0 * dv, 1.0),
        r((size_t)m0 * dv);
    const sdpa_timing keep = E.last;
    const int rc = sdpa_attention_f64(q.data(), k.data(), v.data(), r.data(), m0, n0, dk, dv, flags);
    E.last = keep;
    return rc;
}

// =============================================================================
// device level
// =============================================================================
int sdpa_dev_cvt_d2f(const double *src, float *dst, long rows, int cols, int ld, void *stream) {
    if (rows < 0 || cols <= 0 || check_ld(ld, cols) != SDPA_OK) return SDPA_EINVAL;
    if (rows == 0) return SDPA_OK;
    if (!src || !dst) return SDPA_EINVAL;
    SDPA_TRY(require_device());
    HIP_TRY(sdpa::launch_cvt_d2f(src, dst, rows, cols, ld, (hipStream_t)stream));
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
    if (dv > 1024) return SDPA_EUNSUP;
    SDPA_TRY(require_device());
    PartialArgs a = {};
    a.Q = Qf; a.ldq = ldq; a.K = Kf; a.ldk = ldk; a.V = Vf; a.ldv = ldv;
    a.contrib = contrib; a.ldo = ldo; a.lmax = lmax; a.lsum = lsum;
    a.m = m; a.n_local = n_local; a.dk = dk; a.dv = dv;
    a.kv_splits = sdpa::pick_kv_splits(m, n_local, dk, dv);
    if (a.kv_splits > 1) {
        if (!workspace || workspace_bytes < sdpa::workspace_bytes(m, n_local, dk, dv)) return SDPA_EINVAL;
        a.ws_ld = round4(dv);
        a.ws_contrib = (float *)workspace;
        a.ws_lmax = a.ws_contrib + (size_t)a.kv_splits * m * a.ws_ld;
        a.ws_lsum = a.ws_lmax + (size_t)a.kv_splits * m;
    }
    HIP_TRY(sdpa::launch_shard_partial(a, (hipStream_t)stream));
    return SDPA_OK;
}

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

int sdpa_dev_cvt_d2bf_t(const double *src, void *dst, long rows, int cols, int cols_pad, long ldt,
                        void *stream) {
    if (rows < 0 || cols <= 0 || cols_pad < cols || ldt < rows) return SDPA_EINVAL;
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
