// sdpa_host.hip -- host level of the C ABI: the body of the reference's attention()
// (attention.c:20-75 / attention-mpi.c:191-407) for ONE process driving P ranks
// (P MI355X, or P loopback ranks on one device).
//
//   attention-mpi.c:210-266  K/V convert + Bcast/Scatterv  -> every rank copies ITS rows of K and V
//                                                            straight from the caller's arrays over
//                                                            its own PCIe link, in chunks, and
//                                                            converts on device; the first Q batch
//                                                            starts computing on chunk 0, itself
//                                                            in row pieces as Q arrives
//   attention-mpi.c:268-330  Q ping-pong + MPI_Ibcast      -> two Q slots per rank, copy stream one
//                                                            batch ahead of the compute stream
//                                                            (every rank reads the batch from host
//                                                            memory itself: 8 PCIe links instead of
//                                                            one link + a broadcast)
//   attention-mpi.c:333-338  per-row online softmax         -> the fused kernel, one launch per
//                                                            (row range, K/V chunk); partial triples
//                                                            land in slots and are merged in one pass
//   attention-mpi.c:340-362  Iallreduce MAX / SUM + scales  -> sdpa_coll.h collectives + merge kernels
//   attention-mpi.c:364-399  Ireduce + f2d writeback        -> reduce to rank 0, convert, D2H on the
//                                                            out stream (the last batch in pieces so
//                                                            that only a fraction of it is exposed)
//
// The call only ENQUEUES and waits once at the end; ordering on the devices is by HIP events.  With one
// rank the calling thread enqueues everything.  With P > 1 every rank has its own enqueue thread (a
// pool created with the engine): all ranks' first copies and kernels are issued at the same time, not
// rank after rank; the calling thread page-locks the caller's arrays in the order the copies need them
// and enqueues the per-batch merge collectives on the ranks' COMM streams, so that batch b's
// all-gather / reduce / writeback run under batch b+1's fused kernels (the reference's
// MPI_Ireduce ... next batch ... MPI_Wait, attention-mpi.c:364-380).
#include "sdpa_coll.h"
#include "sdpa_errors.h"
#include "sdpa_hostcvt.h"
#include "sdpa_internal.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <sched.h>
#include <thread>
#include <vector>

namespace {

using sdpa::Bf16Args;
using sdpa::Collectives;
using sdpa::PartialArgs;
using sdpa::RedOp;
using sdpa::round4;

constexpr int kMaxSub = 8;        // row pieces a batch is cut into at the front and at the back of a call

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

struct Rank {
    int dev = 0;                         // HIP device ordinal (loopback ranks share one)
    // s_cp: host->device copies only (never waits for a kernel: PCIe stays busy while the fused
    // kernel holds every CU); s_in: the fp64->operand converts (high priority: they slip into the
    // gap between two fused launches); s_run: fused kernels, merges, collectives; s_out: D2H
    // s_comm: the merge of a batch over the ranks (collectives, merge kernels, fp32->fp64) -- its own
    // stream, so that the NEXT batch's fused kernels on s_run do not queue behind it
    hipStream_t s_cp = nullptr, s_in = nullptr, s_run = nullptr, s_out = nullptr, s_comm = nullptr;
    DevBuf k64, v64;                     // fp64 staging of the WHOLE shard (a copy never waits for a convert)
    DevBuf kf, vf;                       // operand image of the whole shard
    bool vf_view = false;                // fp32: vf is the second half of kf's allocation (V image right behind the K image: one
                                         // pitched copy can then carry a K/V group of the streamed launch into both, stage_chunk)
    DevBuf ws;                           // the fused kernel's own scratch (splits of a direct launch, redo flags)
    DevBuf slots;                        // partial triples of the streamed batch: [slot][row][ldo] + 2 x [slot][row]
    DevBuf q64[2], qf[2], contrib[2], stat[2], gstat[2], red[2], out64[2];
    hipEvent_t ev_q[2] = {}, ev_run[2] = {}, ev_out[2] = {};
    hipEvent_t ev_comm[2] = {};          // the collective tail of the batch in slot s is done with contrib[s]/stat[s]
    hipEvent_t ev_sub[2][kMaxSub] = {};
    hipEvent_t ev_qh[kMaxSub] = {}, ev_qp[kMaxSub] = {};   // Q piece j: copied / converted
    std::vector<hipEvent_t> ev_h2d;      // [2c] K rows, [2c+1] V rows of chunk c have crossed PCIe
    std::vector<hipEvent_t> ev_kv;       // K/V chunk c is converted into the operand image
    std::vector<hipEvent_t> ev_k;        // fused-kernel timing brackets (rank 0)
    hipEvent_t ev_t0 = nullptr, ev_kv_done = nullptr, ev_end = nullptr;
    hipEvent_t ev_tail[4] = {};          // rank 0, last batch's collective tail: start | merged | reduced | widened (comm stream)
    std::vector<hipEvent_t> ev_w;        // host-side widening: piece i of this rank's fp32 result rows has reached the host
    int ev_w_used = 0;
    // the streamed first batch (round 5): ready words on the device (K/V chunks, Q row pieces), the page-locked word the
    // copy engine copies the call's generation from, and the page-locked word a timed-out wait reports in
    unsigned *sflags = nullptr;
    bool sflags_fine = false;
    unsigned *h_gen = nullptr;
    int *h_status = nullptr;
};

// One enqueue thread per rank (P > 1 only).  A job is a function of the rank index; run() hands it to
// every thread and returns, wait() blocks until all of them are done with it.
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    int (*fn)(void *, int) = nullptr;
    void *arg = nullptr;
    unsigned long gen = 0;
    int pending = 0;
    bool stop = false;

    void start(int n, const std::vector<int> &devs) {
        // (a thread only answers to jobs posted after it was created: `gen` survives a shutdown)
        const unsigned long gen0 = gen;
        for (int i = 0; i < n; ++i)
            th.emplace_back([this, i, dev = devs[i], gen0] {
                if (hipSetDevice(dev) != hipSuccess) (void)hipGetLastError();
                unsigned long seen = gen0;
                for (;;) {
                    int (*f)(void *, int);
                    void *a;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_job.wait(lk, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                        f = fn;
                        a = arg;
                    }
                    (void)f(a, i);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
    }
    void run(int (*f)(void *, int), void *a) {
        {
            std::lock_guard<std::mutex> lk(mu);
            fn = f;
            arg = a;
            pending = (int)th.size();
            ++gen;
        }
        cv_job.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_job.notify_all();
        for (std::thread &t : th) t.join();
        th.clear();
        stop = false;
        pending = 0;
    }
};

struct Engine {
    bool up = false;
    int n = 0;                           // ranks
    bool virtual_ranks = false;
    std::vector<Rank> r;
    Collectives *coll = nullptr;
    Pool pool;                           // enqueue threads, one per rank (empty with one rank)
    sdpa::HostConverter *hc = nullptr;   // $SDPA_HOST_CVT=1: fp64 -> operand images on host threads (sdpa_hostcvt.h)
    char *bounce = nullptr;              // 8 KiB page-locked: the partial first / last page of `result` travels through here
    int run_cus = 0;                     // compute units of a rank's compute stream (create_rank)
    int chip_cus = 0;                    // compute units of rank 0's device
    bool rccl_hung = false;              // the last RCCL self-test did not finish (lazy_init does not fall back then)
    unsigned stream_gen = 0x5d000000u;   // generation of the streamed launches' ready words (never 0)
    // Round 6: sticky for the process.  Set when a streamed launch (sdpa_prepare's warm-up call -- the start-up probe -- or a real
    // call) did not see a ready word in time, or when the environment says the copy engines are off (HSA_ENABLE_SDMA=0: every
    // copy is a shader blit, which cannot run beside a launch that owns the chip).  Every later plan takes the launch-per-chunk
    // schedule; the call that found out is re-run on it.  The reference's attention() has no failure mode here
    // (attention-mpi.c:191-407): a drop-in may be slower on a strange runtime, not wrong or dead.
    bool stream_off = false;
    int stream_timeout_ms_once = 0;      // sdpa_prepare's probe: the wait bound of the NEXT call (0 = the default)
    sdpa_timing last = {};
};
// Heap-allocated and never destroyed on purpose: at process exit the order in which this library's
// static destructors and the HIP runtime's run is not ours to choose (inside a Python process the
// runtime belongs to PyTorch), and nothing here needs tearing down then -- sdpa_shutdown() is the
// explicit release.
Engine &E = *new Engine;

double now_us() {
    using namespace std::chrono;
    return duration<double, std::micro>(steady_clock::now().time_since_epoch()).count();
}

int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    const int x = atoi(v);
    return x > 0 ? x : dflt;
}

// Compute units a rank's compute stream leaves to the other streams ($SDPA_COMM_CUS; unset = the default:
// 16 -- two per XCD -- when the call merges over several ranks, so that a batch's collectives and merge
// kernels run UNDER the next batch's fused kernels the way the reference's MPI_Ireduce stays in flight
// (attention-mpi.c:364-380); 0 with one rank, where nothing runs beside the fused kernel).  The fused
// launches size their stream-K grids by what is left, so the reservation costs about its share of the chip
// and no more (round 3: a CU-masked stream broke the exact fit of the grid: +55 %).  Why 16 and not 8
// (measured, one rank with forced collectives at config 4, rocprofv3 kernel trace): with 8 CUs' worth of
// slots free the merge kernel of batch b STARTS beside batch b+1's fused kernel but finishes with it
// (6.9 ms: the dispatcher deals a kernel's workgroups to the shader engines in turn and stalls at the
// first engine without a free slot); with 16 it takes 25-300 us
// (profiles/r04/config4_one_rank_forced_collectives_overlap_reserve{8,16,32}.txt).  A call reserves
// only when it has a NEXT batch to hide a tail under (make_plan: collectives and more than one Q batch).
int comm_cus_reserved(int cus, int ranks) {
    const char *v = getenv("SDPA_COMM_CUS");
    int want = (v && *v) ? atoi(v) : (ranks > 1 ? 16 : 0);
    if (want <= 0) return 0;
    const int xcds = cus >= 64 ? cus / 32 : 1;
    int r = (want + xcds - 1) / xcds * xcds;
    if (r > cus / 2) r = cus / 2 / xcds * xcds;
    return r;
}

// Is this host address page-locked memory?  Known without asking the runtime: the ranges sdpa_host_alloc() handed out
// (what both CLI hosts read the file into).  Any OTHER pointer is taken for pageable and travels through the
// library's page-locked staging -- as fast at every BASELINE shape (profiles/r04/hostlevel_all_configs.log) -- because
// ASKING costs the host application log noise: hipPointerGetAttributes on a plain malloc / numpy array makes ROCm 7
// print "Cannot get amd_mem_obj for ptr" at error level, 3-4 lines per attention() under AMD_LOG_LEVEL >= 1
// (16 KB of the driver's pytest tail was this).  $SDPA_DEBUG host_probe=1 asks after all (a caller that
// page-locks its arrays itself and wants them used in place); the verdict is remembered per base pointer.
struct HostRanges {
    std::mutex mu;
    std::vector<std::pair<const char *, size_t>> r;       // sdpa_host_alloc()ed, not yet freed
    std::vector<std::pair<const void *, bool>> probed;    // $SDPA_DEBUG host_probe=1: base pointer -> page-locked?
};
HostRanges &HR = *new HostRanges;

bool page_locked(const void *p) {
    {
        std::lock_guard<std::mutex> lk(HR.mu);
        for (auto &e : HR.r)
            if ((const char *)p >= e.first && (const char *)p < e.first + e.second) return true;
    }
    static const bool probe = sdpa_debug_int("host_probe", 0) != 0;
    if (!probe) return false;
    {
        std::lock_guard<std::mutex> lk(HR.mu);
        for (auto &e : HR.probed)
            if (e.first == p) return e.second;
    }
    hipPointerAttribute_t at;
    bool locked = false;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) (void)hipGetLastError();
    else locked = at.type == hipMemoryTypeHost;
    std::lock_guard<std::mutex> lk(HR.mu);
    if (HR.probed.size() >= 64) HR.probed.erase(HR.probed.begin());
    HR.probed.push_back({p, locked});
    return locked;
}

// $SDPA_HOST_REGISTER=1: page-lock the caller's pageable arrays for the duration of the call (hipHostRegister), the
// default of rounds 1-3.  OFF by default since round 4: registering and unregistering heap ranges in a process whose
// other code copies from the same addresses with plain (pageable) hipMemcpy -- PyTorch's .cuda() of a numpy array, any
// host application's own copies -- makes the GPU fault on a host page within seconds (the HIP runtime keeps its own
// transient pins of pageable sources by address; tools/gpu_register_stress.py: fault after 5-11 s in every mode that
// registers, 0 faults in 25 000 copies / 6 700 calls when nothing is registered; profiles/r04/).  It was the rare
// silent abort of rounds 2-3.  Pageable arrays now travel through the library's own page-locked staging instead
// (host converts in, host widening out: want_host_cvt / want_host_widen).
bool register_caller_arrays() {
    const char *v = getenv("SDPA_HOST_REGISTER");
    return v && *v && atoi(v) != 0;
}

bool pin_probe() {
    static const bool on = sdpa_debug_int("pin_probe", 1) != 0;
    return on;
}

// RAII page-locking of caller-owned host arrays (best effort: a range that cannot be registered,
// e.g. because the caller already allocated it page-locked, is simply left as it is).
struct HostPins {
    std::vector<void *> ptr;
    double us = 0.0;                                        // host time spent registering
    // ($SDPA_HOST_REGISTER=1 only -- see register_caller_arrays.)  Only the WHOLE PAGES inside [p, p + bytes) are
    // registered: a registration maps its pages into the GPU's address space at their host addresses, and two caller
    // arrays that are neighbours in the heap share the page their boundary falls in.  The partial pages at the two ends
    // are copied as pageable slivers (copy_h2d_cuts / copy_d2h_cuts split there).  (This was round 4's first theory of
    // the suite's rare GPU memory fault; it narrowed nothing -- the fault needs no shared page, only a registration that
    // comes and goes over addresses the runtime has pinned for its own pageable copies:
    // profiles/r04/gpu_memory_fault_root_cause_hipHostRegister.log.)
    static const char *page_up(const void *p) { return (const char *)(((uintptr_t)p + 4095) & ~(uintptr_t)4095); }
    static const char *page_down(const void *p) { return (const char *)((uintptr_t)p & ~(uintptr_t)4095); }
    void add(const void *p, size_t bytes) {
        const char *lo = page_up(p), *hi = page_down((const char *)p + bytes);
        if (hi <= lo || (size_t)(hi - lo) < (1u << 20)) return;   // small ranges: not worth the call
        const double t0 = now_us();
        // Memory the runtime already knows (hipHostMalloc / sdpa_host_alloc arrays, an earlier registration of the
        // caller's, managed memory) is left alone: registering it again is REFUSED by the runtime ("Failed creating
        // memory ... Cannot create memory for size"), and both GPU memory faults of round 4 came 0.5 s and 3.3 s
        // behind exactly two such refusals (tests' sdpa_host_alloc'ed K and V; profiles/r04/).  $SDPA_DEBUG pin_probe=0
        // restores the blind attempt (tools/gpu_register_stress.py compares the two).
        if (pin_probe()) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, lo) != hipSuccess) (void)hipGetLastError();
            else if (at.type != hipMemoryTypeUnregistered) { us += now_us() - t0; return; }
        }
        if (hipHostRegister(const_cast<char *>(lo), (size_t)(hi - lo), hipHostRegisterDefault) == hipSuccess)
            ptr.push_back(const_cast<char *>(lo));
        else
            (void)hipGetLastError();
        us += now_us() - t0;
    }
    ~HostPins() {
        for (void *q : ptr)
            if (hipHostUnregister(q) != hipSuccess) (void)hipGetLastError();
    }
};

// Progressive page-locking registers a caller array in several goes (see sdpa_attention_f64), each a
// range between two page boundaries, so that no two registrations share a page.  A copy that
// straddles such a boundary is issued in parts -- the runtime only treats a host range as
// page-locked when it lies inside ONE registration.  `cuts` = the boundaries inside the array, sorted.
hipError_t copy_h2d_cuts(void *dst, const void *src, size_t bytes, const std::vector<const char *> &cuts,
                         hipStream_t st) {
    const char *s0 = (const char *)src, *end = s0 + bytes;
    char *d0 = (char *)dst;
    for (const char *cut : cuts) {
        if (cut <= s0) continue;
        if (cut >= end) break;
        const size_t head = (size_t)(cut - s0);
        hipError_t e = hipMemcpyAsync(d0, s0, head, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
        d0 += head;
        s0 = cut;
    }
    return hipMemcpyAsync(d0, s0, (size_t)(end - s0), hipMemcpyHostToDevice, st);
}

// the registration boundaries of the call in flight (set before any copy is enqueued, read-only after): inside
// every caller array the first and the last page boundary (the partial pages at its ends are never registered),
// and for K and V the boundaries between the progressive registrations
struct PinCuts {
    std::vector<const char *> k, v, q;
};
PinCuts &CUT = *new PinCuts;

// Whatever path leaves sdpa_attention_f64 -- also an error in the middle of the pipeline -- no
// queued copy, kernel or collective may still reference the caller's arrays when they are
// unregistered and handed back: drain every rank's device first.  (Declared AFTER HostPins so
// that it runs before it.)  A no-op in the success path, which has already waited.
struct DrainOnExit {
    bool armed = true;
    ~DrainOnExit() {
        if (!armed) return;
        for (Rank &r : E.r) {
            if (hipSetDevice(r.dev) != hipSuccess) continue;
            if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
        }
    }
};

// The caller's current device is restored on every exit of a host-level entry point.
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() {
        if (hipGetDevice(&dev) != hipSuccess) {
            (void)hipGetLastError();
            dev = -1;
        }
    }
    ~DeviceRestore() {
        if (dev >= 0 && hipSetDevice(dev) != hipSuccess) (void)hipGetLastError();
    }
};

// ---- the plan of one call ---------------------------------------------------------------
struct Chunk {
    int k0, keys;       // key rows [k0, k0+keys) of the rank's shard
    int splits;         // in-launch K/V splits of the full-row launch
    int slot0;          // first slot it writes
    int group = 0;      // streamed form: the ready word that announces it
    // streamed form, interleaved groups (round 6): the entry is ONE contiguous key range [k0, k0+keys) of the caller's shard that
    // lands as `slices` row ranges of `keys / slices` keys each, `slice_pitch` keys apart, from key row `dst_k0` of the device image
    // (one pitched copy per operand).  slices == 1: a plain range at dst_k0 == k0.
    int dst_k0 = 0, slices = 1, slice_pitch = 0;
    // streamed form, bf16: the entry's columns of the Vt image travel as a PACKED block [padded dv rows][keys_pad] of the host
    // staging, `img_off` elements into it; keys_pad = whole 32-key tiles (the shard's last entry: up to the image's row length)
    long img_off = 0;
    int keys_pad = 0;
};

// The streamed form of a rank's FIRST Q batch (round 5): ONE persistent launch over the whole shard
// (classic grid: query blocks x `splits` equal K/V ranges) that follows its inputs -- instead of one launch per
// (row piece, K/V chunk) with their ramps, tails, slot slabs and 8-16 in-launch splits each.  The shard crosses PCIe
// in GROUPS; group c holds tiles [end_tile[c-1], end_tile[c]) of EVERY split's range (so every workgroup finds its
// next tiles in the next group, whatever split it walks), i.e. `splits` row ranges = `entries`; behind a group's
// copies the copy engine raises the group's ready word.  The triples are those of the device-level launch
// sdpa_dev_shard_partial_f32 on the resident shard, bit for bit.
struct StreamPlan {
    bool on = false;
    // Round 6 (built, measured, OFF by default -- make_plan says why): TWO launches of half the batch's rows
    // each, back to back on the compute stream, each with the split count that fills the chip for ITS rows (twice the one-launch
    // count).  The first half's merge, finish and device-to-host copy then run under the second half's MFMAs instead of behind the
    // one launch -- with one launch nothing leaves early: all its workgroups are resident for its whole length and retire together.
    int halves = 1, rows_per_launch = 0;
    int splits = 1, tiles_per_split = 0;
    std::vector<int> end_tile;          // per group
    std::vector<Chunk> entries;         // staging units in arrival order (group major, split minor)
    // Round 6: INTERLEAVED groups (fp32, shards that are a whole number of tiles per split).  Which keys a split walks is free --
    // softmax(QK^T)V does not care about the order of the keys, only that K row j and V row j stay a pair -- so group c is simply the
    // c-th CONTIGUOUS key range of the caller's shard, and its `splits` equal slices go to tiles [end_tile[c-1], end_tile[c]) of the
    // splits' ranges of the device image: ONE pitched copy per operand and group instead of `splits` row ranges.  The image is then a
    // permutation of the shard (plan_describe: "interleaved": 1; the triples are the device-level launch's on the PERMUTED shard, bit
    // for bit).  With one copy a group the groups can be small: config 2 (8 splits of 1024 keys) crosses PCIe in 4 groups instead of
    // one, and its kernel starts after the first quarter of K/V instead of after all of it.
    bool interleaved = false;
    // ... and where the call is FEED bound (config 2: 0.36 ms of inputs for 0.26 ms of kernel) the Q rows travel as ONE copy between
    // group 0 and group 0's ready word, which then announces both: a ready word is a copy of its own (64 KiB, ~7 us + ~8 us of engine
    // turnaround either side), and no workgroup can do anything before it has its Q rows AND group 0 anyway.
    bool q_with_group0 = false;
};

struct RankPlan {
    int key_off = 0, key_cnt = 0;     // rows of K/V this rank owns (global offsets)
    int row_off = 0, row_cnt = 0;     // query rows this rank computes
    std::vector<Chunk> chunks;        // streaming schedule of the FIRST batch
    int n_slots = 0;                  // total slots of the first batch (1 = direct output)
    size_t ws_bytes = 0;              // scratch of the largest launch
    int max_chunk = 0;
    StreamPlan stream;                // .on: the first batch runs as one streamed launch when the call can feed it
};

struct Plan {
    int m, n, dk, dv;
    bool bf16, qrows, collectives, merge_allreduce;
    bool egress_scatter;              // collectives: reduce-SCATTER the merged contributions, every rank widens and
                                      // sends its rows of the batch home over its own PCIe link;
                                      // false: the reference's reduce to the root (attention-mpi.c:380), $SDPA_EGRESS=root
    int P;
    int cus;                          // compute units the ranks' compute streams may use ($SDPA_COMM_CUS leaves some out)
    int B, nb;                        // rows per Q batch, batches (over the largest row range)
    int row_pieces, piece_min_rows;   // row pieces of the first / last batch (1 = off)
    int ldq, ldk, ldv, ldo;           // leading dimensions of the operand images (elements)
    size_t q_elem, kv_elem;
    std::vector<RankPlan> r;
};

int piece_rows_of(const Plan &pl, int bs) {
    if (pl.row_pieces <= 1 || bs <= 0) return bs > 0 ? bs : 1;
    int pr = (bs + pl.row_pieces - 1) / pl.row_pieces;
    if (pr < pl.piece_min_rows) pr = pl.piece_min_rows;
    pr = (pr + 127) / 128 * 128;
    return pr >= bs ? bs : pr;
}

int pick_splits(const Plan &pl, int rows, int keys) {
    return pl.bf16 ? sdpa::pick_kv_splits_bf16(rows, keys, pl.dk, pl.dv)
                   : sdpa::pick_kv_splits(rows, keys, pl.dk, pl.dv, pl.cus);
}

size_t launch_ws_bytes(const Plan &pl, int rows, int keys) {
    if (pl.bf16) return sdpa_dev_workspace_bytes_bf16(rows, keys, pl.dk, pl.dv);
    return std::max(sdpa::workspace_bytes(rows, keys, pl.dk, pl.dv),
                    sdpa::workspace_bytes_for(rows, pl.dv, sdpa::pick_kv_splits(rows, keys, pl.dk, pl.dv, pl.cus)));
}

// Chunk sizes of a streamed shard: small first (the kernel starts after cmin keys have crossed
// PCIe), doubling up to cmax (long launches run the fused kernel at its best rate).  Boundaries are
// multiples of 1024 keys, which every operand image's tiling divides.
std::vector<int> chunk_sizes(int cnt, int cmin, int cmax) {
    std::vector<int> out;
    if (cnt <= 0) return out;
    if (cnt < 2 * cmin) {
        out.push_back(cnt);
        return out;
    }
    int left = cnt, sz = cmin, at = 0;
    while (left > 0) {
        int take = sz;
        if (left < take + cmin / 2) take = left;       // absorb a short remainder
        out.push_back(take);
        left -= take;
        if (sz > cmin || ++at >= 2) sz = std::min(sz * 2, cmax);   // cmin, cmin, 2cmin, 4cmin, ...
    }
    return out;
}

// Row pieces.  The first batch's Q crosses PCIe in pieces and its FIRST K/V chunk is launched
// piece by piece as they land; the last batch's LAST chunk is launched piece by piece so that the
// merge, finish and D2H of piece j run under the kernel of piece j+1.  Everything in between runs
// full-row launches (the fused kernel's best shape).  A piece is whole query blocks (128 rows).
int piece_rows_of(const struct Plan &pl, int bs);

bool want_bf16(int flags) {
    bool bf16 = (flags & SDPA_F_BF16) != 0;
    if (const char *prec = getenv("SDPA_PRECISION")) bf16 = bf16 || strcmp(prec, "bf16") == 0;
    return bf16;
}

// ... and whether the bf16 kernels take the problem: head dims up to 512 / 1024, and a rank's Vt image (padded dv x
// shard keys x 2 bytes) inside the 32-bit offsets the staging code carries.  Beyond that the call runs the fp32 path
// (and says so on stderr once per call) instead of refusing: fp32 is the tighter of the two tolerances, and the
// reference takes any shape (attention-mpi.c:103-140) -- VERDICT r4 "what's missing" 4.
// (qrows: under the Q-row plan every rank holds ALL n keys -- ADVICE r5: the size test has to know the plan)
bool bf16_for(int flags, int n, int dk, int dv, int ranks, bool qrows) {
    if (!want_bf16(flags)) return false;
    if (dk > 512 || dv > 1024) return false;
    const long per_rank = qrows ? (long)n : ((long)n + ranks - 1) / std::max(1, ranks);
    return (double)sdpa::bf16_pad_dv(dv) * (double)sdpa::bf16_pad_n(per_rank) * 2.0 < 4294967296.0;
}

// ranks > 0: plan for that many ranks without an engine (sdpa_plan_describe: collectives assumed for P > 1)
void make_plan(Plan &pl, int m, int n, int dk, int dv, int flags, int ranks = 0) {
    pl.m = m; pl.n = n; pl.dk = dk; pl.dv = dv;
    pl.P = ranks > 0 ? ranks : E.n;
    pl.qrows = (flags & SDPA_F_PLAN_QROWS) != 0;
    if (const char *v = getenv("SDPA_PLAN")) pl.qrows = pl.qrows || strcmp(v, "qrows") == 0;
    if (pl.P == 1) pl.qrows = false;
    pl.bf16 = bf16_for(flags, n, dk, dv, pl.P, pl.qrows);
    pl.merge_allreduce = (flags & SDPA_F_MERGE_ALLREDUCE) != 0;
    if (const char *v = getenv("SDPA_MERGE")) pl.merge_allreduce = pl.merge_allreduce || strcmp(v, "allreduce") == 0;
    const bool force = sdpa_debug_int("force_collectives", 0) != 0;
    pl.collectives = !pl.qrows && (pl.P > 1 || force) && (ranks > 0 || E.coll != nullptr);
    const char *egress = getenv("SDPA_EGRESS");
    pl.egress_scatter = pl.collectives && pl.P > 1 && !(egress && strcmp(egress, "root") == 0);

    pl.ldo = round4(dv);
    if (pl.bf16) {
        pl.ldq = pl.ldk = sdpa::bf16_pad_dk(dk);
        pl.ldv = 0;
        pl.q_elem = pl.kv_elem = sizeof(unsigned short);
    } else {
        // head dims in (32, 256] are padded to 64 / 128 / 256 columns: every such shape then runs the
        // pipelined LDS-DMA kernel (sdpa_internal.h: dense_ld)
        pl.ldq = pl.ldk = sdpa::dense_ld(dk);
        pl.ldv = sdpa::dense_ld(dv);
        pl.ldo = std::max(pl.ldo, pl.ldv);
        pl.q_elem = pl.kv_elem = sizeof(float);
    }

    const bool no_pipe = (flags & SDPA_F_NO_PIPELINE) != 0;
    // 32768 rows = 256 query blocks: with 2 in-launch splits that is one full wave of 512
    // workgroups, the shape the fused kernel runs fastest at
    int B = env_int("SDPA_QBATCH", 32768);
    // The largest K/V chunk of a streamed shard, when $SDPA_DEBUG kv_chunk_max does not say: how long the inputs take to
    // arrive (fp64 over the link, or through the host's convert pool: ~70 GB/s either way) against how long the kernels
    // take.  Kernel bound (metric shape, configs 3 / 4): data is far ahead of the kernels, a launch costs ~25 us of
    // ramp and tail, so chunks grow to 65536 keys (config 3: 18 -> 8 chunks, 33.3 -> 32.6 ms).  Feed bound (config 5 in
    // bf16: 671 MB in for 4 ms of kernel): what counts is how little kernel is left when the last rows arrive: 8192
    // (8.7 -> 7.8 ms).  profiles/r04/host_sweep_pageable_incremental.log
    int cmax_dflt = 16384;
    {
        const int Pn = ranks > 0 ? ranks : std::max(1, E.n);
        const double keys_r = (double)n / Pn;
        const double t_feed = ((double)m * dk + keys_r * (dk + dv)) * 8.0 / 70e9;
        const double rate = pl.bf16 ? 1.0e15 : (dk <= 256 ? 1.3e14 : 1.0e14);
        const double t_kernel = 2.0 * m * keys_r * (dk + dv) / rate;
        if (t_feed > t_kernel) cmax_dflt = 8192;
        else if (t_feed < 0.5 * t_kernel) cmax_dflt = 65536;
    }
    int cmin = sdpa_debug_pos("kv_chunk_min", 4096), cmax = sdpa_debug_pos("kv_chunk_max", cmax_dflt);
    cmin = std::max(1024, cmin / 1024 * 1024);
    cmax = std::max(cmin, cmax / 1024 * 1024);
    pl.row_pieces = std::min(kMaxSub, sdpa_debug_pos("row_pieces", 4));
    // a piece narrower than 4096 rows runs the fused kernel below ~100 TFLOP/s (tools/gpu_kernel_grid.py)
    pl.piece_min_rows = sdpa_debug_pos("piece_min_rows", 4096);

    pl.r.assign(pl.P, RankPlan());
    int max_rows = 0;
    for (int g = 0; g < pl.P; ++g) {
        RankPlan &rp = pl.r[g];
        if (pl.qrows) {
            rp.key_off = 0; rp.key_cnt = n;
            rp.row_off = sdpa_owner_disp(m, pl.P, g); rp.row_cnt = sdpa_owner_count(m, pl.P, g);
        } else {
            rp.key_off = sdpa_owner_disp(n, pl.P, g); rp.key_cnt = sdpa_owner_count(n, pl.P, g);
            rp.row_off = 0; rp.row_cnt = m;
        }
        max_rows = std::max(max_rows, rp.row_cnt);
    }
    if (no_pipe || B > max_rows) B = max_rows;
    if (B < 1) B = 1;
    pl.B = B;
    pl.nb = (max_rows + B - 1) / B;
    if (pl.nb < 1) pl.nb = 1;
    if (no_pipe) pl.row_pieces = 1;
    // the fused launches leave compute units to the comm streams only when this call has a collective tail AND a next
    // batch whose kernels it can run under; a one-batch call (config 3 at the default batch) gets the whole chip
    {
        const int chip = (ranks > 0 || E.chip_cus <= 0) ? sdpa::kChipCus : E.chip_cus;
        const int reserved = (ranks > 0 || E.run_cus <= 0) ? comm_cus_reserved(chip, pl.P) : chip - E.run_cus;
        // ... and only where it pays: leaving 16 of 256 compute units' worth of slots free costs a batch's kernels
        // 7.6-7.9 % (a 1/2 share of the metric shape 3.93 -> 4.22 ms, a 1/8 share 1.03 -> 1.12:
        // profiles/r04/rank_share_*.json) and hides a tail of ~0.25 ms, i.e. pays below ~3.3 ms of kernel per batch
        // ($SDPA_COMM_CUS set: the caller decides)
        const double rate = pl.bf16 ? 1.0e15 : (dk <= 256 ? 1.3e14 : 1.0e14);
        const double t_batch = 2.0 * B * ((double)n / pl.P) * (dk + dv) / rate;
        const char *forced = getenv("SDPA_COMM_CUS");
        const bool pays = (forced && *forced) || t_batch < 3.3e-3;
        pl.cus = (pl.collectives && pl.nb > 1 && pays) ? chip - reserved : chip;
    }

    for (int g = 0; g < pl.P; ++g) {
        RankPlan &rp = pl.r[g];
        const int rows0 = std::min(B, rp.row_cnt);          // rows of this rank's first batch
        std::vector<int> sizes = no_pipe ? std::vector<int>(rp.key_cnt > 0 ? 1 : 0, rp.key_cnt)
                                         : chunk_sizes(rp.key_cnt, cmin, cmax);
        // the first chunk is launched in row pieces, and so is the last one when the first batch is
        // also the last and finishes its rows itself (collectives run once per batch)
        const bool first_is_last = rp.row_cnt <= B;
        const int rows_piece0 = piece_rows_of(pl, rows0);
        int k0 = 0, slot = 0;
        for (size_t ci = 0; ci < sizes.size(); ++ci) {
            const int sz = sizes[ci];
            Chunk c;
            c.k0 = k0; c.keys = sz;
            const bool in_pieces = ci == 0 || (ci + 1 == sizes.size() && first_is_last && !pl.collectives);
            const int launch_rows = in_pieces ? rows_piece0 : rows0;
            c.splits = launch_rows > 0 ? pick_splits(pl, launch_rows, sz) : 1;
            c.slot0 = slot;
            slot += c.splits;
            k0 += sz;
            rp.chunks.push_back(c);
            rp.max_chunk = std::max(rp.max_chunk, sz);
        }
        rp.n_slots = slot;
        // scratch: streamed launches carry their slots themselves and need only the bf16 redo
        // flags from ws; direct launches (later batches, pieces of the last one) need their splits
        size_t ws = 0;
        const int nb_g = rp.row_cnt > 0 ? (rp.row_cnt + B - 1) / B : 0;
        const int rows_last = rp.row_cnt - (nb_g - 1) * B;                  // rows of this rank's last batch
        const int pr = piece_rows_of(pl, rows_last);
        const int shapes[6] = {rows0, rows_last, pr, pr > 0 ? rows_last % pr : 0, rows_piece0,
                               rows_piece0 > 0 ? rows0 % rows_piece0 : 0};
        for (int rows : shapes)
            if (rows > 0 && rp.key_cnt > 0) {
                ws = std::max(ws, launch_ws_bytes(pl, rows, rp.key_cnt));
                for (const Chunk &c : rp.chunks) ws = std::max(ws, launch_ws_bytes(pl, rows, c.keys));
            }
        rp.ws_bytes = ws;

        // ---- the streamed form of the first batch (StreamPlan): shape conditions only; whether the call can FEED it
        //      (host converts into page-locked staging: no kernel may have to run beside the persistent launch) is the
        //      call's decision (sdpa_attention_f64).  Classic grids only: a launch that stream-K would cut differently
        //      (a reservation, an odd m) keeps the chunked schedule.
        rp.stream = StreamPlan();
        const char *sv = getenv("SDPA_STREAMED");
        const bool stream_knob = (!(sv && *sv) || atoi(sv) != 0) && !E.stream_off;
        const int scmin = std::max(1024, sdpa_debug_pos("stream_chunk_min", cmin) / 1024 * 1024);
        // bf16 (round 5, second half): the tandem kernel's shapes (value columns in 512-wide chunks) have a persistent form too;
        // its K groups are row ranges of the bf16 image, its V groups COLUMN ranges of the Vt image, written by the host
        // (sdpa_hostcvt: submit_t) and carried by pitched copies -- the copy engine's as well (profiles/r05/copy_engine_probes.log).
        // A whole-chip launch only: the bf16 kernels have no stream-K form to size for a reservation.
        const bool whole_chip = pl.cus >= ((ranks > 0 || E.chip_cus <= 0) ? sdpa::kChipCus : E.chip_cus);
        const bool bf16_streamable = pl.bf16 && whole_chip && sdpa::bf16_stream_launch_supported(dk, dv);
        if (stream_knob && (!pl.bf16 || bf16_streamable) && !no_pipe && (pl.bf16 || sdpa::stream_launch_supported(dk, dv)) && rows0 > 0 &&
            rp.key_cnt >= 2 * scmin) {
            sdpa::F32Plan fp = {};
            auto plan_rows = [&](int rows) {
                sdpa::F32Plan q = {};
                if (pl.bf16) {
                    q.splits = sdpa::pick_kv_splits_bf16(rows, rp.key_cnt, dk, dv);
                    q.streamk = 0;
                } else {
                    q = sdpa::plan_f32_launch(rows, rp.key_cnt, dk, dv, pl.cus);
                }
                return q;
            };
            fp = plan_rows(rows0);
            // two half-row launches where the rank sends its rows home itself, the row pieces pair up, each half still fills the
            // chip within the streamed form's 8 splits, and a launch is long enough to hide a half's egress under (>= ~1.5 ms)
            int halves = 1, rows_launch = rows0;
            {
                const int pieces0 = rows_piece0 > 0 ? (rows0 + rows_piece0 - 1) / rows_piece0 : 1;
                const double rate = pl.bf16 ? 1.0e15 : (dk <= 256 ? 1.3e14 : 1.0e14);
                const double t_launch = 2.0 * rows0 * (double)rp.key_cnt * (dk + dv) / rate;
                // OFF by default -- a measured negative result (profiles/r06/two_wave_egress_ab.log, same box): the tail shrinks as
                // intended (headline 0.50 -> 0.32 ms, config 5 in bf16 1.50 -> 0.89) but the launches lose more than that: twice the
                // split slabs and a second ramp at the headline (kernels 8.22 -> 8.42 ms, call 9.01 -> 9.04), and where the call is LINK
                // bound the first half cannot end before the last K/V group has landed, so the second half's MFMAs come BEHIND the
                // transfer instead of under it (config 5 bf16: kernels 4.5 -> 5.7 ms, call 6.10 -> 6.73; config 3: 32.0 -> 32.8).
                // $SDPA_DEBUG=two_wave=1 runs it (tests/test_gpu_host_pipeline.py keeps it bit-identical to the device-level halves).
                const bool knob = sdpa_debug_int("two_wave", 0) != 0;
                if (knob && !pl.collectives && first_is_last && pieces0 >= 2 && pieces0 % 2 == 0 && rows0 % rows_piece0 == 0 && t_launch >= 3.0e-3) {
                    const int half = pieces0 / 2 * rows_piece0;
                    const sdpa::F32Plan h = plan_rows(half);
                    if (!h.streamk && h.splits <= 8 && rows0 == 2 * half) {
                        halves = 2;
                        rows_launch = half;
                        fp = h;
                    }
                }
            }
            const long nqb = (rows0 + sdpa::kQRowsPerBlock - 1) / sdpa::kQRowsPerBlock;
            // (at most 8 splits: every group crosses PCIe as `splits` row ranges of K and of V, and below ~256 KiB a
            //  copy costs more to enqueue than to move -- few query blocks keep the launch-per-chunk schedule)
            (void)nqb;   // (a grid of more than one round is fine: later rounds find their words raised)
            if (!fp.streamk && fp.splits <= 8) {
                StreamPlan &sp = rp.stream;
                sp.halves = halves;
                sp.rows_per_launch = rows_launch;
                sp.splits = fp.splits;
                const int ntiles = (rp.key_cnt + sdpa::kKvTile - 1) / sdpa::kKvTile;
                sp.tiles_per_split = (ntiles + sp.splits - 1) / sp.splits;
                // (at most kStreamMaxChunks groups: a long shard with a small largest chunk -- feed bound plans -- gets larger ones)
                const int gmax = std::max(std::max(scmin, cmax), (int)(((long)rp.key_cnt / 12 + 1023) / 1024 * 1024));
                const std::vector<int> gsz = chunk_sizes(rp.key_cnt, scmin, gmax);
                // A group crosses PCIe as one row range per split and operand, and a copy costs ~10 us whatever its size:
                // below ~2048 keys of a split per group (1 MiB at d = 128) the copies, not the link, set the pace (config 2
                // with 2 groups x 8 splits: 32 copies of 256 KiB, 1.02-1.05 ms against 0.99 chunked -- profiles/r05/
                // boundary_streamed_vs_chunked_ab.log).  So a split's share of a group is at least $SDPA_DEBUG stream_entry_min
                // keys; a shard too short for two such groups is ONE group -- whose ranges are adjacent: one copy.
                int entry_min = std::max(1, sdpa_debug_pos("stream_entry_min", 2048) / sdpa::kKvTile);
                std::vector<int> gsz_i;
                const bool interleave = !pl.bf16 && sp.splits > 1 && sdpa_debug_int("stream_interleave", 1) != 0 &&
                                        (long)sp.splits * sp.tiles_per_split * sdpa::kKvTile == (long)rp.key_cnt;
                if (interleave) {
                    // one pitched copy per group and operand whatever the split count: a GROUP (not a split's share of it) is at
                    // least stream_entry_min keys, and the groups may be as small as that
                    entry_min = std::max(1, entry_min / sp.splits);
                    const int smin = std::max(1024, sdpa_debug_pos("stream_interleave_min", 2048) / 1024 * 1024);
                    // feed bound (the inputs take longer to arrive than the kernel takes: config 2): what counts is how little work is
                    // left when the LAST group lands -- equal small groups; kernel bound: start early, then grow (fewer waits)
                    const double t_feed_r = ((double)rows0 * dk + (double)rp.key_cnt * (dk + dv)) * 8.0 / 70e9;
                    const double t_kern_r = 2.0 * rows0 * (double)rp.key_cnt * (dk + dv) / (dk <= 256 ? 1.3e14 : 1.0e14);
                    const int smax = sdpa_debug_pos("stream_interleave_max", t_feed_r > t_kern_r ? smin : gmax);
                    gsz_i = chunk_sizes(rp.key_cnt, std::min(scmin, smin), std::max(std::min(scmin, smin), std::max(smax, (int)(((long)rp.key_cnt / 12 + 1023) / 1024 * 1024))));
                    // (forced at the metric shape -- kernel bound -- it LOSES: 8.82 8.84 -> 9.02 9.00 ms; the row pieces' early starts matter there)
                    sp.q_with_group0 = t_feed_r > t_kern_r && halves == 1 && sdpa_debug_int("stream_q_with_group0", 1) != 0;
                }
                const std::vector<int> &gsz_use = interleave ? gsz_i : gsz;
                long cum = 0;
                int prev = 0;
                for (size_t gi = 0; gi < gsz_use.size(); ++gi) {
                    cum += gsz_use[gi];
                    int end = (int)(((double)cum / rp.key_cnt) * sp.tiles_per_split + 0.5);
                    end = std::max(end, prev + entry_min);
                    if (gi + 1 == gsz_use.size() || sp.tiles_per_split - end < entry_min) end = sp.tiles_per_split;
                    end = std::min(sp.tiles_per_split, end);
                    if (end > prev) sp.end_tile.push_back(end);
                    prev = end;
                    if (prev >= sp.tiles_per_split) break;
                }
                if (!sp.end_tile.empty() && sp.end_tile.back() < sp.tiles_per_split) sp.end_tile.back() = sp.tiles_per_split;
                sp.interleaved = interleave && sp.end_tile.size() >= 2 && (int)sp.end_tile.size() <= sdpa::kStreamMaxChunks;
                sp.q_with_group0 = sp.q_with_group0 && sp.interleaved;
                if (interleave && !sp.interleaved && sp.end_tile.size() > 1) sp.end_tile.assign(1, sp.tiles_per_split);
                if (sp.interleaved) {
                    for (size_t gi = 0; gi < sp.end_tile.size(); ++gi) {
                        const int t0 = gi ? sp.end_tile[gi - 1] : 0, t1 = sp.end_tile[gi];
                        Chunk e;
                        e.k0 = sp.splits * t0 * sdpa::kKvTile; e.keys = sp.splits * (t1 - t0) * sdpa::kKvTile;
                        e.splits = sp.splits; e.slot0 = -1; e.group = (int)gi;
                        e.dst_k0 = t0 * sdpa::kKvTile; e.slices = sp.splits; e.slice_pitch = sp.tiles_per_split * sdpa::kKvTile;
                        sp.entries.push_back(e);
                    }
                }
                if (sp.end_tile.size() >= 1 && (int)sp.end_tile.size() <= sdpa::kStreamMaxChunks) {
                    for (size_t gi = 0; gi < sp.end_tile.size() && !sp.interleaved; ++gi)
                        for (int sx = 0; sx < sp.splits; ++sx) {
                            const long t0 = gi ? sp.end_tile[gi - 1] : 0, t1 = sp.end_tile[gi];
                            const long k0 = ((long)sx * sp.tiles_per_split + t0) * sdpa::kKvTile;
                            const long k1 = std::min<long>(((long)sx * sp.tiles_per_split + t1) * sdpa::kKvTile, rp.key_cnt);
                            if (k1 <= k0) continue;
                            if (!sp.entries.empty() && sp.entries.back().group == (int)gi &&
                                sp.entries.back().k0 + sp.entries.back().keys == (int)k0) {
                                sp.entries.back().keys += (int)(k1 - k0);        // adjacent ranges of one group: one copy
                                continue;
                            }
                            Chunk e;
                            e.k0 = (int)k0; e.keys = (int)(k1 - k0); e.splits = sp.splits; e.slot0 = -1; e.group = (int)gi;
                            e.dst_k0 = e.k0;
                            sp.entries.push_back(e);
                        }
                    if (pl.bf16) {
                        const long ldn = sdpa::bf16_pad_n(rp.key_cnt), rows_img = sdpa::bf16_pad_dv(dv);
                        long off = 0;
                        for (Chunk &e : sp.entries) {
                            const bool last = e.k0 + e.keys == rp.key_cnt;
                            e.keys_pad = (int)(last ? ldn - e.k0 : (e.keys + 31) / 32 * 32);
                            e.img_off = off;
                            off += rows_img * e.keys_pad;
                        }
                    }
                    sp.on = true;
                    if (!pl.bf16) rp.ws_bytes = std::max(rp.ws_bytes, sdpa::workspace_bytes_for(rows_launch, dv, sp.splits));
                    else rp.ws_bytes = std::max(rp.ws_bytes, launch_ws_bytes(pl, rows_launch, rp.key_cnt));
                    // (bf16: the launch's own scratch -- launch_ws_bytes(rows0, key_cnt) above -- holds its splits and redo flags)
                }
            }
        }
    }
}

int ensure_buffers(const Plan &pl) {
    for (int g = 0; g < pl.P; ++g) {
        Rank &rk = E.r[g];
        const RankPlan &rp = pl.r[g];
        HIP_TRY(hipSetDevice(rk.dev));
        SDPA_TRY(ensure(rk.k64, (size_t)rp.key_cnt * pl.dk * sizeof(double)));
        SDPA_TRY(ensure(rk.v64, (size_t)rp.key_cnt * pl.dv * sizeof(double)));
        // (bf16: whole 32-key tiles of K rows -- the tiled image's last tile is read to its end, sdpa_internal.h)
        if (pl.bf16) {
            if (rk.vf_view) { rk.vf = DevBuf(); rk.vf_view = false; }
            SDPA_TRY(ensure(rk.kf, (size_t)sdpa::bf16_pad_n(rp.key_cnt) * pl.ldk * pl.kv_elem));
            SDPA_TRY(ensure(rk.vf, (size_t)sdpa::bf16_pad_dv(pl.dv) * sdpa::bf16_pad_n(rp.key_cnt) * sizeof(unsigned short)));
        } else {            // fp32: ONE allocation, the V image right behind the K image
            const size_t kb = (size_t)rp.key_cnt * pl.ldk * sizeof(float), vb = (size_t)rp.key_cnt * pl.ldv * sizeof(float);
            if (!rk.vf_view && rk.vf.p) { HIP_TRY(hipFree(rk.vf.p)); rk.vf = DevBuf(); }
            SDPA_TRY(ensure(rk.kf, kb + vb));
            rk.vf.p = (char *)rk.kf.p + kb;
            rk.vf.cap = 0;
            rk.vf_view = true;
        }
        SDPA_TRY(ensure(rk.ws, rp.ws_bytes));
        const size_t B = (size_t)pl.B;
        if (rp.n_slots > 1) SDPA_TRY(ensure(rk.slots, (size_t)rp.n_slots * B * (pl.ldo + 2) * sizeof(float)));
        // rows of a batch a rank sends home: all of them (it finishes its own rows, or it is the root of
        // the reduce), or its 1/P share of the reduce-scatter
        const size_t share = (B + pl.P - 1) / pl.P;
        const size_t out_rows = !pl.collectives ? B : pl.egress_scatter ? share : (g == 0 ? B : 0);
        for (int s = 0; s < 2; ++s) {
            SDPA_TRY(ensure(rk.q64[s], B * pl.dk * sizeof(double)));
            SDPA_TRY(ensure(rk.qf[s], B * pl.ldq * pl.q_elem));
            // (+P rows: the reduce-scatter sends P equal shares, the last ones padded past the batch)
            SDPA_TRY(ensure(rk.contrib[s], (B + pl.P) * pl.ldo * sizeof(float)));
            SDPA_TRY(ensure(rk.stat[s], 2 * B * sizeof(float)));
            if (pl.collectives) {
                SDPA_TRY(ensure(rk.gstat[s], (pl.merge_allreduce ? 2 : 2 * (size_t)pl.P) * B * sizeof(float)));
                if (out_rows) SDPA_TRY(ensure(rk.red[s], out_rows * pl.ldo * sizeof(float)));
            }
            if (out_rows) SDPA_TRY(ensure(rk.out64[s], out_rows * pl.dv * sizeof(double)));
        }
        if (rp.stream.on && !rk.sflags) {
            const size_t words = (size_t)(sdpa::kStreamMaxChunks + sdpa::kStreamMaxPieces) * sdpa::kStreamFlagStride + 16;    // (+ the abort word)
            rk.sflags_fine = hipExtMallocWithFlags((void **)&rk.sflags, words * sizeof(unsigned), hipDeviceMallocFinegrained) == hipSuccess;
            if (!rk.sflags_fine) {
                (void)hipGetLastError();
                HIP_TRY(hipMalloc((void **)&rk.sflags, words * sizeof(unsigned)));
            }
            HIP_TRY(hipMemset(rk.sflags, 0, words * sizeof(unsigned)));
            HIP_TRY(hipHostMalloc((void **)&rk.h_gen, sdpa::kStreamFlagStride * sizeof(unsigned), hipHostMallocPortable));
            HIP_TRY(hipHostMalloc((void **)&rk.h_status, 64, hipHostMallocPortable));
            memset(rk.h_gen, 0, sdpa::kStreamFlagStride * sizeof(unsigned));
            *rk.h_status = 0;
        }
        while (rk.ev_kv.size() < std::max(rp.chunks.size(), rp.stream.entries.size()) + 1) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            rk.ev_kv.push_back(e);
            for (int h = 0; h < 2; ++h) {
                HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                rk.ev_h2d.push_back(e);
            }
        }
    }
    return SDPA_OK;
}

// ---- one fused launch ---------------------------------------------------------------------
// Rows [j0, j0+jr) of batch slot s against keys [k0, k0+keys) of the rank's shard.  slot0 < 0:
// output straight into contrib[s]/stat[s] (the launch merges its own splits); otherwise the
// `splits` partial triples go to slots [slot0, slot0+splits) of the streamed batch (bs rows).
int launch_fused(const Plan &pl, Rank &rk, const RankPlan &rp, int s, int bs, int j0, int jr, int k0,
                 int keys, int splits, int slot0) {
    float *contrib = (float *)rk.contrib[s].p + (size_t)j0 * pl.ldo;
    float *lmax = (float *)rk.stat[s].p + j0;
    float *lsum = (float *)rk.stat[s].p + bs + j0;
    float *sl_c = nullptr, *sl_m = nullptr, *sl_s = nullptr;
    if (slot0 >= 0) {
        float *base = (float *)rk.slots.p;
        float *mbase = base + (size_t)rp.n_slots * bs * pl.ldo;
        float *sbase = mbase + (size_t)rp.n_slots * bs;
        sl_c = base + ((size_t)slot0 * bs + j0) * pl.ldo;
        sl_m = mbase + (size_t)slot0 * bs + j0;
        sl_s = sbase + (size_t)slot0 * bs + j0;
    }
    if (keys <= 0 || !pl.bf16) {
        // (an empty shard also takes this launcher in bf16 mode: its T = 0 path writes the
        //  (0, -inf, 0) triple of attention-mpi.c:172-173 and never touches K or V)
        PartialArgs a = {};
        const bool dummy = keys <= 0 && pl.bf16;
        a.Q = (const float *)rk.qf[s].p + (dummy ? 0 : (size_t)j0 * pl.ldq);
        a.ldq = dummy ? 4 : pl.ldq;
        a.K = (const float *)rk.kf.p + (dummy ? 0 : (size_t)k0 * pl.ldk);
        a.ldk = dummy ? 4 : pl.ldk;
        a.V = (const float *)rk.vf.p + (dummy ? 0 : (size_t)k0 * pl.ldv);
        a.ldv = dummy ? 4 : pl.ldv;
        a.m = jr; a.n_local = keys > 0 ? keys : 0; a.dk = dummy ? 4 : pl.dk; a.dv = pl.dv;
        a.kv_splits = keys > 0 ? splits : 1;
        a.cus = pl.cus;
        if (slot0 >= 0 && a.kv_splits == 1) {
            a.contrib = sl_c; a.ldo = pl.ldo; a.lmax = sl_m; a.lsum = sl_s;
        } else {
            a.contrib = contrib; a.ldo = pl.ldo; a.lmax = lmax; a.lsum = lsum;
            if (slot0 >= 0) {
                a.ws_contrib = sl_c; a.ws_lmax = sl_m; a.ws_lsum = sl_s;
                a.ws_ld = pl.ldo; a.ws_rows = bs; a.defer_merge = 1;
            } else if (a.kv_splits > 1) {
                sdpa::carve_workspace(a, rk.ws.p, pl.ldo);
            }
        }
        HIP_TRY(sdpa::launch_shard_partial(a, rk.s_run));
        return SDPA_OK;
    }
    Bf16Args a = {};
    a.Q = (const unsigned short *)rk.qf[s].p + (size_t)j0 * pl.ldq;  a.ldq = pl.ldq;
    a.K = (const unsigned short *)rk.kf.p + (size_t)k0 * pl.ldk;     a.ldk = pl.ldk;
    a.Vt = (const unsigned short *)rk.vf.p + sdpa::bf16_vt_key_offset(k0, pl.dv);   a.ldvt = sdpa::bf16_pad_n(rp.key_cnt);
    a.m = jr; a.n_local = keys; a.dk = pl.dk; a.dv = pl.dv;
    a.kv_splits = splits;
    if (slot0 >= 0 && splits == 1) {
        a.contrib = sl_c; a.ldo = pl.ldo; a.lmax = sl_m; a.lsum = sl_s;
        sdpa::bf16_carve_workspace(a, rk.ws.p, pl.ldo);          // redo flags only
    } else if (slot0 >= 0) {
        a.contrib = contrib; a.ldo = pl.ldo; a.lmax = lmax; a.lsum = lsum;
        a.ws_contrib = sl_c; a.ws_lmax = sl_m; a.ws_lsum = sl_s;
        a.ws_ld = pl.ldo; a.ws_rows = bs; a.defer_merge = 1;
        a.redo = sdpa::bf16_needs_redo(pl.dk, pl.dv) ? (int *)rk.ws.p : nullptr;
    } else {
        a.contrib = contrib; a.ldo = pl.ldo; a.lmax = lmax; a.lsum = lsum;
        sdpa::bf16_carve_workspace(a, rk.ws.p, pl.ldo);
    }
    HIP_TRY(sdpa::launch_shard_partial_bf16(a, rk.s_run));
    return SDPA_OK;
}

// Merge the n_slots partial triples of rows [j0, j0+jr) into contrib[s]/stat[s] (the reference's
// merge algebra, attention-mpi.c:340-351, applied to the chunks and splits of one rank).
int merge_slots(const Plan &pl, Rank &rk, const RankPlan &rp, int s, int bs, int j0, int jr) {
    PartialArgs a = {};
    float *base = (float *)rk.slots.p;
    float *mbase = base + (size_t)rp.n_slots * bs * pl.ldo;
    float *sbase = mbase + (size_t)rp.n_slots * bs;
    a.m = jr; a.dv = pl.dv; a.kv_splits = rp.n_slots;
    a.ws_contrib = base + (size_t)j0 * pl.ldo; a.ws_ld = pl.ldo; a.ws_rows = bs;
    a.ws_lmax = mbase + j0; a.ws_lsum = sbase + j0;
    a.contrib = (float *)rk.contrib[s].p + (size_t)j0 * pl.ldo; a.ldo = pl.ldo;
    a.lmax = (float *)rk.stat[s].p + j0;
    a.lsum = (float *)rk.stat[s].p + bs + j0;
    HIP_TRY(sdpa::launch_split_merge(a, rk.s_run));
    return SDPA_OK;
}

// K rows (is_v = false) or V rows (true) of chunk c of the rank's shard: host -> device into the
// fp64 staging image on the copy stream, then the convert into the operand image
// (attention-mpi.c:224-225 / :248-249 and the Scatterv of :258-264) on the convert stream.
// $SDPA_HOST_CVT: the operand images of the call in flight as host threads write them (page-locked
// staging, whole-problem layout: row r of K / V / Q at r * ld elements) and the conversion task that
// fills each piece; a copy waits for its task on the host, then moves the finished image rows.
struct HostImages {
    sdpa::HostConverter *cv = nullptr;
    char *k = nullptr, *v = nullptr, *q = nullptr;
    int ldv_host = 0;                                      // row stride of the host V image (bf16: dense dv)
    std::vector<size_t> v_rank_off;                        // bf16, streamed: byte offset of rank g's packed Vt image in `v`
    std::vector<char> streamed;                            // [rank]: chunk indices below name rp.stream.entries, not rp.chunks
    bool pair = false;                                     // fp32, every rank streamed with interleaved groups, K and V images equally wide: `k` holds
                                                           // group after group [K rows][V rows] (the group of shard keys [k0, k0+keys) at row 2*k0), and
                                                           // ONE pitched copy carries a group into both device images (stage_chunk)
    std::vector<std::vector<int>> k_task, v_task;          // [rank][chunk]
    std::vector<std::vector<std::vector<int>>> q_task;     // [rank][batch][row piece]
};
HostImages &HI = *new HostImages;                          // (valid while a call with host converts is in flight)

int stage_half(const Plan &pl, Rank &rk, const RankPlan &rp, const double *src, int c, bool is_v, int g = -1) {
    // `bare` (a streamed rank): copies only -- NO event is recorded on the copy stream and nothing waits for one.  An event
    // record is a packet in the stream's hardware queue, and a process has few of those (4 per priority): the copy stream
    // may share one with a compute stream whose persistent launch is waiting for these very bytes -- the packet, and every
    // copy behind it, would sit behind that launch (round 5, call 4: two loopback ranks = 5 streams on 4 queues, rank 0's
    // group 1 never arrived).  Copies of this size are the copy engine's and order themselves by the stream alone.
    const bool bare = g >= 0 && g < (int)HI.streamed.size() && HI.streamed[g];
    const Chunk &ch = bare ? rp.stream.entries[c] : rp.chunks[c];
    const size_t row0 = (size_t)rp.key_off + ch.k0;
    const int cols = is_v ? pl.dv : pl.dk;
    hipEvent_t copied = rk.ev_h2d[2 * c + (is_v ? 1 : 0)];
    if (HI.cv && g >= 0) {
        // the host threads have written (or are writing) this chunk's rows of the operand image
        HI.cv->wait(is_v ? HI.v_task[g][c] : HI.k_task[g][c]);
        if (!is_v || !pl.bf16) {
            const int ld = is_v ? pl.ldv : pl.ldk;
            const size_t el = is_v ? sizeof(float) : pl.kv_elem;
            if (bare && ch.slices > 1) {       // an interleaved group: its slices land in the splits' ranges -- ONE pitched copy
                const size_t slice = (size_t)(ch.keys / ch.slices) * ld * el;
                HIP_TRY(hipMemcpy2DAsync((char *)(is_v ? rk.vf.p : rk.kf.p) + (size_t)ch.dst_k0 * ld * el, (size_t)ch.slice_pitch * ld * el,
                                         (is_v ? HI.v : HI.k) + row0 * ld * el, slice, slice, (size_t)ch.slices, hipMemcpyHostToDevice, rk.s_cp));
                return SDPA_OK;
            }
            char *img = (char *)(is_v ? rk.vf.p : rk.kf.p) + (size_t)ch.k0 * ld * el;
            HIP_TRY(hipMemcpyAsync(img, (is_v ? HI.v : HI.k) + row0 * ld * el, (size_t)ch.keys * ld * el,
                                   hipMemcpyHostToDevice, rk.s_cp));
            if (bare) return SDPA_OK;
            HIP_TRY(hipEventRecord(copied, rk.s_cp));
            HIP_TRY(hipStreamWaitEvent(rk.s_in, copied, 0));
            return SDPA_OK;
        }
        if (bare) {
            // streamed: the host wrote the entry's columns of the Vt image (packed: keys_pad elements a row); a pitched copy puts
            // them in place -- the copy engine's, like the linear ones (tools/probes/h2d_2d_probe.hip)
            const long ldn = sdpa::bf16_pad_n(rp.key_cnt);
            const unsigned short *img = (const unsigned short *)(HI.v + HI.v_rank_off[g]) + ch.img_off;
            if (sdpa::bf16_tiled(pl.dv)) {      // tiled image (round 6): the entry's tiles are ONE contiguous block at both ends
                HIP_TRY(hipMemcpyAsync((unsigned short *)rk.vf.p + sdpa::bf16_vt_key_offset(ch.k0, pl.dv), img,
                                       (size_t)ch.keys_pad * sdpa::bf16_pad_dv(pl.dv) * sizeof(unsigned short), hipMemcpyHostToDevice, rk.s_cp));
                return SDPA_OK;
            }
            HIP_TRY(hipMemcpy2DAsync((unsigned short *)rk.vf.p + ch.k0, (size_t)ldn * sizeof(unsigned short), img,
                                     (size_t)ch.keys_pad * sizeof(unsigned short), (size_t)ch.keys_pad * sizeof(unsigned short),
                                     (size_t)sdpa::bf16_pad_dv(pl.dv), hipMemcpyHostToDevice, rk.s_cp));
            return SDPA_OK;
        }
        // bf16 V: the rows travel as dense bf16, the device transposes them into the Vt image
        unsigned short *rows16 = (unsigned short *)rk.v64.p + (size_t)ch.k0 * pl.dv;
        HIP_TRY(hipMemcpyAsync(rows16, HI.v + row0 * pl.dv * sizeof(unsigned short),
                               (size_t)ch.keys * pl.dv * sizeof(unsigned short), hipMemcpyHostToDevice, rk.s_cp));
        HIP_TRY(hipEventRecord(copied, rk.s_cp));
        HIP_TRY(hipStreamWaitEvent(rk.s_in, copied, 0));
        const long ldn = sdpa::bf16_pad_n(rp.key_cnt);
        const bool last = ch.k0 + ch.keys == rp.key_cnt;
        const long pad = last ? ldn - ch.k0 : ch.keys;
        HIP_TRY(sdpa::launch_cvt_bf_t_part(rows16, (unsigned short *)rk.vf.p + sdpa::bf16_vt_key_offset(ch.k0, pl.dv), ch.keys, pad, pl.dv,
                                           sdpa::bf16_pad_dv(pl.dv), ldn, rk.s_in));
        return SDPA_OK;
    }
    double *stage = (double *)(is_v ? rk.v64.p : rk.k64.p) + (size_t)ch.k0 * cols;
    HIP_TRY(copy_h2d_cuts(stage, src + row0 * cols, (size_t)ch.keys * cols * sizeof(double), is_v ? CUT.v : CUT.k,
                          rk.s_cp));
    HIP_TRY(hipEventRecord(copied, rk.s_cp));
    HIP_TRY(hipStreamWaitEvent(rk.s_in, copied, 0));
    if (!is_v) {
        if (pl.bf16)       // (the image of the shape's kernel: tiled for dv > 256, with the last tile's pad rows zeroed)
            HIP_TRY(sdpa::launch_cvt_d2bf_k(stage, (unsigned short *)rk.kf.p + (size_t)ch.k0 * pl.ldk, ch.keys,
                                            ch.k0 + ch.keys == rp.key_cnt ? sdpa::bf16_pad_n(rp.key_cnt) - ch.k0 : ch.keys, pl.dk, pl.dv,
                                            rk.s_in));
        else
            HIP_TRY(sdpa::launch_cvt_d2f(stage, (float *)rk.kf.p + (size_t)ch.k0 * pl.ldk, ch.keys, pl.dk, pl.ldk,
                                         rk.s_in));
    } else if (pl.bf16) {
        const long ldn = sdpa::bf16_pad_n(rp.key_cnt);
        const bool last = ch.k0 + ch.keys == rp.key_cnt;
        const long pad = last ? ldn - ch.k0 : ch.keys;       // the image's zero tail belongs to the last keys
        HIP_TRY(sdpa::launch_cvt_d2bf_t_part(stage, (unsigned short *)rk.vf.p + sdpa::bf16_vt_key_offset(ch.k0, pl.dv), ch.keys, pad, pl.dv,
                                             sdpa::bf16_pad_dv(pl.dv), ldn, rk.s_in));
    } else {
        HIP_TRY(sdpa::launch_cvt_d2f(stage, (float *)rk.vf.p + (size_t)ch.k0 * pl.ldv, ch.keys, pl.dv, pl.ldv,
                                     rk.s_in));
    }
    return SDPA_OK;
}

// What sdpa_kv_prefetch() has already put on the devices for the NEXT compute call.
struct Prefetched {
    bool active = false;
    const double *K = nullptr, *V = nullptr;
    int m = 0, n = 0, dk = 0, dv = 0, flags = 0;
    std::vector<std::vector<char>> k_done, v_done;     // [rank][chunk]
    void reset() { *this = Prefetched(); }
    bool matches(const double *k, const double *v, int m_, int n_, int dk_, int dv_, int flags_) const {
        return active && K == k && V == v && m == m_ && n == n_ && dk == dk_ && dv == dv_ && flags == flags_;
    }
};
Prefetched &PF = *new Prefetched;

// Both halves of chunk c (skipping what a prefetch already moved), then the chunk's ready event.
int stage_chunk(const Plan &pl, Rank &rk, const RankPlan &rp, int g, const double *K, const double *V, int c) {
    if (HI.cv && HI.pair && g >= 0 && g < (int)HI.streamed.size() && HI.streamed[g]) {
        // an interleaved group of the streamed launch, K and V rows side by side in the staging: its `slices` K slices go to the
        // splits' ranges of the K image, its V slices to the same places of the V image, which lies right behind (ensure_buffers):
        // 2 * slices rows of ONE pitched copy (a copy costs ~8 us of engine turnaround whatever its size: profiles/r06/
        // config2_boundary_timeline_*.txt)
        const Chunk &ch = rp.stream.entries[c];
        const size_t rowb = (size_t)pl.ldk * sizeof(float), slice = (size_t)(ch.keys / ch.slices) * rowb;
        char *dst = (char *)rk.kf.p + (size_t)ch.dst_k0 * rowb;
        const char *src = HI.k + 2 * ((size_t)rp.key_off + ch.k0) * rowb;
        const size_t pitch = (size_t)ch.slice_pitch * rowb;
        HI.cv->wait(HI.k_task[g][c]);
        if (ch.group == 0) {      // the first group: its K half leaves as soon as it is converted (the converters have only just woken up)
            HIP_TRY(hipMemcpy2DAsync(dst, pitch, src, slice, slice, (size_t)ch.slices, hipMemcpyHostToDevice, rk.s_cp));
            HI.cv->wait(HI.v_task[g][c]);
            HIP_TRY(hipMemcpy2DAsync(dst + ch.slices * pitch, pitch, src + ch.slices * slice, slice, slice, (size_t)ch.slices, hipMemcpyHostToDevice, rk.s_cp));
            return SDPA_OK;
        }
        HI.cv->wait(HI.v_task[g][c]);
        HIP_TRY(hipMemcpy2DAsync(dst, pitch, src, slice, slice, 2 * (size_t)ch.slices, hipMemcpyHostToDevice, rk.s_cp));
        return SDPA_OK;
    }
    const bool have_k = PF.active && PF.k_done[g][c], have_v = PF.active && PF.v_done[g][c];
    if (!have_k) SDPA_TRY(stage_half(pl, rk, rp, K, c, false, g));
    if (!have_v) SDPA_TRY(stage_half(pl, rk, rp, V, c, true, g));
    if (g >= 0 && g < (int)HI.streamed.size() && HI.streamed[g]) return SDPA_OK;     // (bare copies: see stage_half)
    HIP_TRY(hipEventRecord(rk.ev_kv[c], rk.s_in));
    return SDPA_OK;
}

// Rows [j0, j0+jr) of a Q batch: copy, convert into slot s of qf (attention-mpi.c:303,:325).
int stage_q_rows(const Plan &pl, Rank &rk, const double *Q, int s, size_t i0, int j0, int jr, hipEvent_t copied,
                 hipEvent_t converted, int task = -1, bool bare = false) {
    if (HI.cv && task >= 0) {
        HI.cv->wait(task);
        HIP_TRY(hipMemcpyAsync((char *)rk.qf[s].p + (size_t)j0 * pl.ldq * pl.q_elem, HI.q + (i0 + j0) * pl.ldq * pl.q_elem,
                               (size_t)jr * pl.ldq * pl.q_elem, hipMemcpyHostToDevice, rk.s_cp));
        if (bare) return SDPA_OK;                 // (a streamed rank: copies only, stage_half says why)
        HIP_TRY(hipEventRecord(copied, rk.s_cp));
        HIP_TRY(hipStreamWaitEvent(rk.s_in, copied, 0));
        HIP_TRY(hipEventRecord(converted, rk.s_in));
        return SDPA_OK;
    }
    double *q64 = (double *)rk.q64[s].p + (size_t)j0 * pl.dk;
    HIP_TRY(copy_h2d_cuts(q64, Q + (i0 + j0) * pl.dk, (size_t)jr * pl.dk * sizeof(double), CUT.q, rk.s_cp));
    HIP_TRY(hipEventRecord(copied, rk.s_cp));
    HIP_TRY(hipStreamWaitEvent(rk.s_in, copied, 0));
    if (pl.bf16)
        HIP_TRY(sdpa::launch_cvt_d2bf_q(q64, (unsigned short *)rk.qf[s].p + (size_t)j0 * pl.ldq, jr, pl.dk, pl.ldq, rk.s_in));
    else
        HIP_TRY(sdpa::launch_cvt_d2f(q64, (float *)rk.qf[s].p + (size_t)j0 * pl.ldq, jr, pl.dk, pl.ldq, rk.s_in));
    HIP_TRY(hipEventRecord(converted, rk.s_in));
    return SDPA_OK;
}

int coll_fail() {
    fprintf(stderr, "sdpa: collective failed (%s): %s\n", E.coll ? E.coll->name() : "?",
            E.coll ? E.coll->last_error() : "");
    return SDPA_ERCCL;
}

// internal: a streamed launch gave up waiting for a ready word (never returned to the caller: sdpa_attention_f64 re-runs the call)
constexpr int kStreamTimedOut = -1000;

// ---- one sdpa_attention_f64 call ---------------------------------------------------------------
void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

struct Call {
    const double *Q = nullptr, *K = nullptr, *V = nullptr;
    double *result = nullptr;
    int m = 0, n = 0, dk = 0, dv = 0;
    Plan pl;
    double t_enter = 0.0;
    HostPins pins;
    bool do_pin = true, progressive = false, threaded = false;
    bool hostcvt = false;              // K, V, Q are read by the host's convert threads: only `result` is page-locked
    bool streamed = false;             // ranks whose plan allows it run their first batch as ONE streamed launch (StreamPlan)
    unsigned stream_gen = 0;           // ... whose ready words carry this generation
    unsigned long long stream_timeout_ticks = 0;
    int stream_timeout_ms = 0;
    int stream_drop_word = -1;         // $SDPA_STREAM_DROP_WORD (tests only): this ready word is never raised -> the launch must time out, not hang
    size_t k_bytes = 0, v_bytes = 0;
    // progressive page-locking: what the first copies of every rank need (stage 0) and the rest of K/V (stage 2)
    std::vector<std::pair<const char *, size_t>> pin0, pin2;
    std::atomic<int> pin_done{-1};                 // highest registration stage completed (0..3)
    std::atomic<int> enq[sdpa::kMaxRanks];         // batches rank g has completely enqueued
    std::atomic<int> tails{0};                     // batches whose collective tail is enqueued
    std::atomic<int> failed{0};                    // first error of any thread
    double first_kernel_us[sdpa::kMaxRanks] = {};  // entry -> rank g's first fused launch enqueued (host clock)
    int n_brackets = 0, last_splits = 1;           // rank 0's enqueue thread only
    int last_rows = 0, last_keys = 0;              // shape of rank 0's last fused launch
    sdpa::LaunchNote last_note = {};               // ... and what it was (recorded by the launcher on that thread)
    bool tail_marked = false;                      // the last batch's collective tail recorded root.ev_tail[0..3]
    // bytes of `result` that lie in its partial first / last page: copied into E.bounce by the device, into place after the wait
    struct Sliver { char *dst; const char *src; size_t bytes; };
    std::vector<Sliver> slivers;
    std::mutex sliver_mu;
    // host-side widening ($SDPA_HOST_WIDEN): result rows come home as fp32 into page-locked staging `w_base` (row i of the
    // result at w_base + i*dv floats) and host threads widen them into `result` (which is then never registered)
    bool widen = false;
    float *w_base = nullptr;
    struct WidenPiece { hipEvent_t landed; size_t row0; int rows; };
    std::vector<WidenPiece> w_pieces;             // in enqueue order (guarded by sliver_mu)
    double widen_us = 0.0;                        // host time inside the widening calls
    Call() { for (auto &e : enq) e.store(0); }
    int fail(int code) {
        int none = 0;
        failed.compare_exchange_strong(none, code);
        return code;
    }
};

// Finished rows -> the caller's `result`.  The whole pages inside `result` are registered (true asynchronous DMA); the
// partial page at its start and at its end is not (HostPins::add), and a device-to-host copy into pageable memory
// would BLOCK the enqueuing thread until the rows exist.  Those few bytes (< 4 KiB each) go into a page-locked bounce
// buffer instead and are put in place by the host after the call's one wait.
int copy_result_rows(Call &c, double *dst_rows, const void *src, size_t bytes, hipStream_t st) {
    char *d0 = (char *)dst_rows, *end = d0 + bytes;
    const char *s0 = (const char *)src;
    char *r0 = (char *)c.result, *r1 = r0 + (size_t)c.m * c.dv * sizeof(double);
    char *a0 = (char *)HostPins::page_up(r0), *a1 = (char *)HostPins::page_down(r1);
    if (!c.do_pin || !E.bounce || a1 <= a0) {                 // nothing is registered: one plain copy
        HIP_TRY(hipMemcpyAsync(d0, s0, bytes, hipMemcpyDeviceToHost, st));
        return SDPA_OK;
    }
    auto sliver = [&](char *lo, char *hi, char *edge_base, int slot) -> int {     // [lo, hi) inside the partial page
        if (hi <= lo) return SDPA_OK;
        char *b = E.bounce + slot * 4096 + (lo - edge_base);
        HIP_TRY(hipMemcpyAsync(b, s0 + (lo - d0), (size_t)(hi - lo), hipMemcpyDeviceToHost, st));
        std::lock_guard<std::mutex> lk(c.sliver_mu);
        c.slivers.push_back({lo, b, (size_t)(hi - lo)});
        return SDPA_OK;
    };
    SDPA_TRY(sliver(d0, std::min(end, a0), r0, 0));                            // head: [r0, a0)
    char *m0 = std::max(d0, a0), *m1 = std::min(end, a1);
    if (m1 > m0) HIP_TRY(hipMemcpyAsync(m0, s0 + (m0 - d0), (size_t)(m1 - m0), hipMemcpyDeviceToHost, st));
    SDPA_TRY(sliver(std::max(d0, a1), end, a1, 1));                            // tail: [a1, r1)
    return SDPA_OK;
}

// Host-side widening (attention-mpi.c:373 / :396: the ROOT widens the reduced rows with cvt_f2d_avx512): rows
// [row0, row0 + rows) of the result exist as fp32 on rank rk's device -- `src32`, `ld` floats a row, already normalised
// when lsum == nullptr, else still to be divided by lsum (merge step 5, :358-362).  They are made dense in `dense32`
// (device), cross PCIe as fp32 -- half the bytes of the fp64 rows -- into the page-locked staging, and the calling
// thread hands them to the converter pool once `landed` has fired (widen_landed_pieces).
int ship_rows_f32(Call &c, Rank &rk, size_t row0, int rows, const float *src32, int ld, const float *lsum,
                  float *dense32, hipStream_t produced_on, hipEvent_t produced, hipStream_t st) {
    if (rows <= 0) return SDPA_OK;
    const int dv = c.dv;
    const float *from = src32;
    if (lsum || ld != dv) {
        HIP_TRY(sdpa::launch_finish_f32(src32, ld, lsum, dense32, rows, dv, produced_on));
        from = dense32;
    }
    if (st != produced_on) {
        HIP_TRY(hipEventRecord(produced, produced_on));
        HIP_TRY(hipStreamWaitEvent(st, produced, 0));
    }
    HIP_TRY(hipMemcpyAsync(c.w_base + row0 * dv, from, (size_t)rows * dv * sizeof(float), hipMemcpyDeviceToHost, st));
    if (rk.ev_w_used == (int)rk.ev_w.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        rk.ev_w.push_back(e);
    }
    hipEvent_t landed = rk.ev_w[rk.ev_w_used++];
    HIP_TRY(hipEventRecord(landed, st));
    std::lock_guard<std::mutex> lk(c.sliver_mu);
    c.w_pieces.push_back({landed, row0, rows});
    return SDPA_OK;
}

// the calling thread, behind the enqueue of the whole call: piece after piece, wait until the fp32 rows are in the
// staging area (a spin on the event: the call blocks anyway, and the last piece's latency is the call's tail),
// then widen them into `result` on the pool's threads and this one
int widen_landed_pieces(Call &c) {
    for (size_t i = 0; i < c.w_pieces.size(); ++i) {
        const Call::WidenPiece &w = c.w_pieces[i];
        for (;;) {
            const hipError_t e = hipEventQuery(w.landed);
            if (e == hipSuccess) break;
            if (e != hipErrorNotReady) HIP_TRY(e);
            (void)hipGetLastError();
            cpu_relax();
        }
        const double t0 = now_us();
        E.hc->widen(c.w_base + w.row0 * c.dv, c.result + w.row0 * c.dv, (size_t)w.rows * c.dv);
        c.widen_us += now_us() - t0;
    }
    return SDPA_OK;
}

// Page-lock the caller's arrays for the duration of the call.  Copies from pages the driver has never
// seen run at ~11 GB/s on this platform (tools/probes/h2d_probe.cpp); registered ones at ~57 GB/s and
// truly asynchronously, which the enqueue-then-wait structure relies on for its overlap (it stays
// correct without).  Registering costs ~3.8 us per MiB of host time, so it is done PROGRESSIVELY, in
// the order the copies need it: stage 0 = the first K/V chunk of every rank (page-aligned hulls),
// 1 = Q, 2 = the K/V remainders, 3 = `result` (whose pages a caller has typically never touched:
// registering faults them in).  One rank: the calling thread registers each stage right behind the
// enqueue of the work it does not gate -- the GPU works while the host registers.  P ranks: the calling
// thread registers stage after stage while the ranks' threads enqueue; a thread waits for a stage only
// in front of the copies that need it.  Nothing stays registered after the call.
void pin_stage(Call &c, int stage) {
    if (c.do_pin) {
        const size_t res_bytes = c.widen ? 0 : (size_t)c.m * c.dv * sizeof(double);   // widened by the host: no DMA into it
        if (c.hostcvt) {
            if (stage == 3) c.pins.add(c.result, res_bytes);
        } else if (!c.progressive) {
            if (stage == 0) {
                c.pins.add(c.K, c.k_bytes);
                c.pins.add(c.V, c.v_bytes);
                c.pins.add(c.Q, (size_t)c.m * c.dk * sizeof(double));
                c.pins.add(c.result, res_bytes);
            }
        } else {
            switch (stage) {
                case 0: for (auto &r : c.pin0) c.pins.add(r.first, r.second); break;
                case 1: c.pins.add(c.Q, (size_t)c.m * c.dk * sizeof(double)); break;
                case 2: for (auto &r : c.pin2) c.pins.add(r.first, r.second); break;
                default: c.pins.add(c.result, res_bytes); break;
            }
        }
    }
    c.pin_done.store(stage, std::memory_order_release);
}

// in front of the copies that need registration stage `stage`
void need_pin(Call &c, int stage) {
    if (!c.threaded) {
        for (int st = c.pin_done.load(std::memory_order_relaxed) + 1; st <= stage; ++st) pin_stage(c, st);
        return;
    }
    while (c.pin_done.load(std::memory_order_acquire) < stage && !c.failed.load(std::memory_order_relaxed)) cpu_relax();
}

// The registration plan of a K/V-sharded call: per array, the page-aligned hull of every rank's first
// chunk (stage 0), the gaps between the hulls (stage 2) and the boundaries a copy must be split at.
bool plan_progressive_pins(Call &c) {
    const Plan &pl = c.pl;
    if (pl.qrows) return false;
    auto one = [&](const double *base, int cols, size_t total, std::vector<const char *> &cuts) -> bool {
        const uintptr_t b0 = (uintptr_t)base, end = b0 + total;
        uintptr_t prev_hi = b0;
        bool first = true;
        for (int g = 0; g < pl.P; ++g) {
            const RankPlan &rp = pl.r[g];
            if (rp.key_cnt <= 0) continue;
            if (rp.chunks.size() < 2) return false;            // nothing to stream behind: register up front
            const size_t row = (size_t)cols * sizeof(double);
            uintptr_t lo = first ? b0 : ((b0 + (size_t)rp.key_off * row) & ~(uintptr_t)4095);
            uintptr_t hi = (b0 + ((size_t)rp.key_off + rp.chunks[0].keys) * row + 4095) & ~(uintptr_t)4095;
            if (hi > end) hi = end;
            if (lo < prev_hi) lo = prev_hi;
            if (hi <= lo) return false;
            if (lo > prev_hi) c.pin2.push_back({(const char *)prev_hi, (size_t)(lo - prev_hi)});
            c.pin0.push_back({(const char *)lo, (size_t)(hi - lo)});
            if (lo > b0) cuts.push_back((const char *)lo);
            if (hi < end) cuts.push_back((const char *)hi);
            prev_hi = hi;
            first = false;
        }
        if (first) return false;
        if (prev_hi < end) c.pin2.push_back({(const char *)prev_hi, (size_t)(end - prev_hi)});
        return true;
    };
    std::vector<const char *> kc, vc;
    if (!one(c.K, c.dk, c.k_bytes, kc) || !one(c.V, c.dv, c.v_bytes, vc)) {
        c.pin0.clear();
        c.pin2.clear();
        return false;
    }
    CUT.k = kc;
    CUT.v = vc;
    return true;
}

// Rows [j0, j0+jr) of the batch in slot s are complete in contrib[s]/stat[s]: step 5 with gsum = lsum fused with the
// writeback (attention-mpi.c:358-362, :373), then they go home -- as fp32 rows that the host widens, or as fp64 rows.
// `finished`: the dense rows are in out64[s] already (the launch's fused merge + finish pass wrote them): only the copy home is left.
int finish_rows(Call &c, int g, int s, int bs, size_t i0, int ev, int j0, int jr, bool finished = false, bool in_order = false) {
    const Plan &pl = c.pl;
    Rank &rk = E.r[g];
    const int dv = c.dv;
    need_pin(c, 3);                  // behind the enqueue of (nearly) all of the batch's kernels
    if (c.widen) {
        if (finished)   // (in_order: nothing is left to run under the copy -- it goes into the compute stream itself, behind the pass that
                        //  wrote the rows, and starts without the ~20 us a cross-stream event costs)
            return ship_rows_f32(c, rk, i0 + j0, jr, (const float *)rk.out64[s].p + (size_t)j0 * dv, dv, nullptr,
                                 (float *)rk.out64[s].p + (size_t)j0 * dv, rk.s_run, rk.ev_sub[s][ev], in_order ? rk.s_run : rk.s_out);
        return ship_rows_f32(c, rk, i0 + j0, jr, (const float *)rk.contrib[s].p + (size_t)j0 * pl.ldo, pl.ldo,
                             (const float *)rk.stat[s].p + bs + j0, (float *)rk.out64[s].p + (size_t)j0 * dv,
                             rk.s_run, rk.ev_sub[s][ev], rk.s_out);
    }
    if (!finished)
        HIP_TRY(sdpa::launch_finish_f64((const float *)rk.contrib[s].p + (size_t)j0 * pl.ldo, pl.ldo,
                                    (const float *)rk.stat[s].p + bs + j0,
                                    (double *)rk.out64[s].p + (size_t)j0 * dv, jr, dv, rk.s_run));
    HIP_TRY(hipEventRecord(rk.ev_sub[s][ev], rk.s_run));
    HIP_TRY(hipStreamWaitEvent(rk.s_out, rk.ev_sub[s][ev], 0));
    SDPA_TRY(copy_result_rows(c, c.result + (i0 + j0) * dv, (double *)rk.out64[s].p + (size_t)j0 * dv,
                              (size_t)jr * dv * sizeof(double), rk.s_out));
    return SDPA_OK;
}

// The first Q batch of rank g as ONE persistent launch that follows its inputs (StreamPlan; round 5).  Order of the
// enqueue: the launch goes out FIRST -- it waits by itself, in the kernel, for the ready word of whatever it is about to
// read -- then the copy stream is fed in the order the workgroups want the bytes: group 0 of the shard (the head of every
// split's range), the Q row pieces, the other groups; behind each the copy engine raises its ready word (a 64 KiB copy
// of the call's generation: no kernel is involved, none could run while the launch holds every workgroup slot -- which
// is why it is not a 4-byte copy: this runtime hands those to a shader, sdpa_internal.h).
// Nothing on the device side waits for the host beyond that: a host that is slow to convert simply keeps workgroups
// waiting (bounded: $SDPA_STREAM_TIMEOUT_MS, then the call fails instead of hanging).
int rank_batch0_streamed(Call &c, int g) {
    const Plan &pl = c.pl;
    Rank &rk = E.r[g];
    Rank &root = E.r[0];
    const RankPlan &rp = pl.r[g];
    const StreamPlan &sp = rp.stream;
    const int s = 0;
    const int bs = std::min(pl.B, rp.row_cnt);
    const size_t i0 = (size_t)rp.row_off;
    HIP_TRY(hipSetDevice(rk.dev));
    const bool finisher = !pl.collectives;
    const int pr = piece_rows_of(pl, bs);
    const int pieces = (bs + pr - 1) / pr;
    if (pieces > sdpa::kStreamMaxPieces || (pieces > 1 && pr % sdpa::kQRowsPerBlock != 0)) return SDPA_EINVAL;
    for (int i = 0; i < sdpa::kStreamFlagStride; ++i) rk.h_gen[i] = c.stream_gen;     // (64 KiB: a few microseconds)
    *rk.h_status = 0;

    // ---- 1. the launch(es): ONE over the batch's rows, or (StreamPlan::halves == 2) one per half -- the second right behind the
    //      first half's merge and finish kernels on the compute stream, so that the first half's rows cross PCIe and are widened
    //      while the second half computes.  Both follow the same K/V groups; a half waits for ITS Q row pieces.
    const int halves = sp.halves == 2 && finisher && pieces % 2 == 0 && sp.rows_per_launch * 2 == bs ? 2 : 1;
    const int rows_launch = halves == 2 ? sp.rows_per_launch : bs;
    const int pieces_launch = pieces / halves;
    auto bracket = [&]() -> int {              // timing event on rank 0's compute stream
        if (g != 0) return SDPA_OK;
        if ((int)root.ev_k.size() <= c.n_brackets) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            root.ev_k.push_back(e);
        }
        HIP_TRY(hipEventRecord(root.ev_k[c.n_brackets++], root.s_run));
        return SDPA_OK;
    };
    sdpa::StreamArgs st = {};
    st.flags = rk.sflags;
    st.gen = c.stream_gen;
    st.n_chunks = (int)sp.end_tile.size();
    for (int i = 0; i < st.n_chunks; ++i) st.chunk_end[i] = sp.end_tile[i];
    st.q_piece_blocks = std::max(1, (pr + sdpa::kQRowsPerBlock - 1) / sdpa::kQRowsPerBlock);
    const bool q_with_group0 = sp.q_with_group0 && halves == 1 && HI.cv != nullptr;
    if (q_with_group0) st.q_piece_blocks = 0;          // (no Q word: group 0's word announces the Q rows too)
    st.timeout_ticks = c.stream_timeout_ticks;
    st.status = rk.h_status;
    st.abort = rk.sflags + (size_t)(sdpa::kStreamMaxChunks + sdpa::kStreamMaxPieces) * sdpa::kStreamFlagStride;
    // a rank that finishes its own rows: the launch's merge pass also normalises and writes the dense rows that go home (round 6:
    // split_merge_finish_kernel -- the finish kernels of the row pieces and a round trip through contrib are gone)
    const bool fused_finish = finisher && sp.splits > 1;
    // (sending the rows home from that pass itself -- streaming stores into the page-locked staging, no device-to-host copies -- was
    //  built and measured in round 6: stores from a kernel cross PCIe at 36-40 GB/s against the copies' 55; config 2 0.76 -> 0.76-0.79 ms,
    //  the metric shape 8.92-8.97 -> 8.94-8.95: not kept, profiles/r06/config2_boundary_steps.log)
    auto fin_of = [&](size_t r0) -> sdpa::FinishTarget {
        sdpa::FinishTarget f = {nullptr, nullptr};
        if (c.widen) f.out32 = (float *)rk.out64[s].p + r0 * c.dv;
        else f.out64 = (double *)rk.out64[s].p + r0 * c.dv;
        return f;
    };
    auto launch_half = [&](int h) -> int {
        const size_t r0 = (size_t)h * rows_launch;
        st.q_piece0 = h * pieces_launch;
        SDPA_TRY(bracket());
        if (pl.bf16) {
            Bf16Args b = {};
            b.Q = (const unsigned short *)rk.qf[s].p + r0 * pl.ldq;  b.ldq = pl.ldq;
            b.K = (const unsigned short *)rk.kf.p;     b.ldk = pl.ldk;
            b.Vt = (const unsigned short *)rk.vf.p;    b.ldvt = sdpa::bf16_pad_n(rp.key_cnt);
            b.m = rows_launch; b.n_local = rp.key_cnt; b.dk = pl.dk; b.dv = pl.dv;
            b.kv_splits = sp.splits;
            b.contrib = (float *)rk.contrib[s].p + r0 * pl.ldo; b.ldo = pl.ldo;
            b.lmax = (float *)rk.stat[s].p + r0;
            b.lsum = (float *)rk.stat[s].p + bs + r0;
            sdpa::bf16_carve_workspace(b, rk.ws.p, pl.ldo);
            b.defer_merge = fused_finish ? 1 : 0;
            HIP_TRY(sdpa::launch_shard_partial_bf16_streamed(b, st, rk.s_run));
            if (fused_finish) {
                PartialArgs p = {};
                p.lmax = b.lmax; p.lsum = b.lsum; p.m = b.m; p.dv = b.dv; p.kv_splits = b.kv_splits;
                p.ws_contrib = b.ws_contrib; p.ws_ld = b.ws_ld; p.ws_lmax = b.ws_lmax; p.ws_lsum = b.ws_lsum; p.ws_rows = b.m;
                HIP_TRY(sdpa::launch_split_merge_finish(p, fin_of(r0), rk.s_run));
            }
        } else {
            PartialArgs a = {};
            a.Q = (const float *)rk.qf[s].p + r0 * pl.ldq; a.ldq = pl.ldq;
            a.K = (const float *)rk.kf.p;    a.ldk = pl.ldk;
            a.V = (const float *)rk.vf.p;    a.ldv = pl.ldv;
            a.m = rows_launch; a.n_local = rp.key_cnt; a.dk = pl.dk; a.dv = pl.dv;
            a.kv_splits = sp.splits;
            a.cus = pl.cus;
            a.contrib = (float *)rk.contrib[s].p + r0 * pl.ldo; a.ldo = pl.ldo;
            a.lmax = (float *)rk.stat[s].p + r0;
            a.lsum = (float *)rk.stat[s].p + bs + r0;
            if (a.kv_splits > 1) sdpa::carve_workspace(a, rk.ws.p, pl.ldo);
            a.defer_merge = fused_finish ? 1 : 0;
            HIP_TRY(sdpa::launch_shard_partial_streamed(a, st, rk.s_run));
            if (fused_finish) {
                a.defer_merge = 0;
                HIP_TRY(sdpa::launch_split_merge_finish(a, fin_of(r0), rk.s_run));
            }
        }
        SDPA_TRY(bracket());
        return SDPA_OK;
    };
    SDPA_TRY(launch_half(0));
    if (c.first_kernel_us[g] == 0.0) c.first_kernel_us[g] = now_us() - c.t_enter;
    if (g == 0) {
        c.last_splits = sp.splits;
        c.last_rows = rows_launch;
        c.last_keys = rp.key_cnt;
        c.last_note = sdpa::last_launch_note();
    }

    // ---- 2. its inputs, in the order it wants them
    auto raise = [&](int word) -> int {
        if (word == c.stream_drop_word) return SDPA_OK;
        // (a whole 64 KiB block of generation words: a copy of that size is the copy engine's, a 4-byte one a shader's)
        HIP_TRY(hipMemcpyAsync((void *)(rk.sflags + (size_t)word * sdpa::kStreamFlagStride), rk.h_gen,
                               sdpa::kStreamFlagStride * sizeof(unsigned), hipMemcpyHostToDevice, rk.s_cp));
        return SDPA_OK;
    };
    const int E_n = (int)sp.entries.size();
    int e = 0;
    auto stage_group = [&](int gi, bool announce = true) -> int {
        for (; e < E_n && sp.entries[e].group == gi; ++e) SDPA_TRY(stage_chunk(pl, rk, rp, g, c.K, c.V, e));
        return announce ? raise(gi) : SDPA_OK;
    };
    need_pin(c, 0);
    SDPA_TRY(stage_group(0, !q_with_group0));
    need_pin(c, 1);
    if (q_with_group0) {          // group 0, the batch's Q rows as ONE copy, then the word that announces both
        for (int j = 0; j < pieces; ++j) HI.cv->wait(HI.q_task[g][0][j]);
        HIP_TRY(hipMemcpyAsync(rk.qf[s].p, HI.q + i0 * pl.ldq * pl.q_elem, (size_t)bs * pl.ldq * pl.q_elem, hipMemcpyHostToDevice, rk.s_cp));
        SDPA_TRY(raise(0));
    }
    for (int j = 0; j < pieces && !q_with_group0; ++j) {
        SDPA_TRY(stage_q_rows(pl, rk, c.Q, s, i0, j * pr, std::min(pr, bs - j * pr), rk.ev_qh[j], rk.ev_qp[j],
                              HI.q_task[g][0][j], true));
        SDPA_TRY(raise(sdpa::kStreamMaxChunks + j));
    }
    need_pin(c, 2);
    for (int gi = 1; gi < st.n_chunks; ++gi) SDPA_TRY(stage_group(gi));
    // the only packets this rank puts into the copy stream's queue: BEHIND the last ready word (if they wait for the
    // launch in a shared queue, nothing the launch needs waits with them).  ev_q[s] = "slot s's Q image has been written"
    // for batch 2's staging (which also waits for ev_run[s], i.e. for this launch); ev_kv_done = rank 0's timing mark.
    HIP_TRY(hipEventRecord(rk.ev_q[s], rk.s_cp));
    if (g == 0) HIP_TRY(hipEventRecord(rk.ev_kv_done, rk.s_cp));

    if (halves == 2) {
        // the first half's rows: finished on the compute stream BEHIND its launch and IN FRONT of the second one (a finish kernel
        // could not become resident beside a launch that owns the chip), shipped on the egress stream under the second launch.
        // Enqueued only NOW, behind every copy of the call: the egress stream's wait for the finish kernel is a packet in a
        // hardware queue the copy stream may share (5 streams, 4 queues) -- in front of the copies it would hold back the very
        // bytes the first launch is waiting for.  The first launch cannot end before its last group has been enqueued, so the
        // second one is never late.
        for (int j = 0; j < pieces_launch; ++j) SDPA_TRY(finish_rows(c, g, s, bs, i0, j, j * pr, std::min(pr, bs - j * pr), fused_finish));
        SDPA_TRY(launch_half(1));
    }

    // ---- 3. its rows: merged by the launcher's split-merge pass; a rank that finishes its own rows sends them home
    //      in row pieces (finish + D2H of piece j under the host's widening of piece j-1)
    const bool egress_in_order = fused_finish && pl.nb == 1 && sdpa_debug_int("egress_in_order", 1) != 0;
    if (finisher)
        for (int j = halves == 2 ? pieces_launch : 0; j < pieces; ++j)
            SDPA_TRY(finish_rows(c, g, s, bs, i0, j, j * pr, std::min(pr, bs - j * pr), fused_finish, egress_in_order));
    HIP_TRY(hipEventRecord(rk.ev_run[s], rk.s_run));
    if (finisher) HIP_TRY(hipEventRecord(rk.ev_out[s], rk.s_out));
    return SDPA_OK;
}

// Everything rank g enqueues for Q batch b: its inputs (K/V chunks with the first batch), the fused
// launches, and -- when it finishes its rows itself (no merge collective) -- finish + D2H.
// host converts: chunks staged ahead of the launch that is being enqueued.  One is enough -- chunk ch+1's copy then
// runs under chunk ch's kernel, and the wait for chunk ch+2's conversion happens under it too -- and with chunk sizes
// that double, every further chunk of look-ahead holds a launch back until twice as many rows are converted
constexpr int kStageAhead = 1;

int rank_batch(Call &c, int g, int b) {
    const Plan &pl = c.pl;
    Rank &rk = E.r[g];
    Rank &root = E.r[0];
    const RankPlan &rp = pl.r[g];
    const int s = b & 1;
    const int j_lo = b * pl.B;
    if (j_lo >= rp.row_cnt) return SDPA_OK;                  // this rank has no rows left
    if (b == 0 && c.streamed && rp.stream.on) return rank_batch0_streamed(c, g);
    const int bs = std::min(pl.B, rp.row_cnt - j_lo);
    const size_t i0 = (size_t)rp.row_off + j_lo;             // first global query row
    const int C = (int)rp.chunks.size();
    HIP_TRY(hipSetDevice(rk.dev));

    auto bracket = [&]() -> int {              // timing event on rank 0's compute stream
        if (g != 0) return SDPA_OK;
        if ((int)root.ev_k.size() <= c.n_brackets) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            root.ev_k.push_back(e);
        }
        HIP_TRY(hipEventRecord(root.ev_k[c.n_brackets++], root.s_run));
        return SDPA_OK;
    };

    const bool finisher = !pl.collectives;               // finishes its own rows on this rank
    const bool last_batch = j_lo + pl.B >= rp.row_cnt;   // of this rank
    const bool head_pieces = b == 0;                     // Q arrives in row pieces
    const bool tail_pieces = last_batch && finisher;     // rows leave in row pieces
    const int pr = (head_pieces || tail_pieces) ? piece_rows_of(pl, bs) : bs;
    const int pieces = (bs + pr - 1) / pr;

    // copy + convert streams: K/V chunk 0, then the Q batch (in pieces for batch 0).  q64[s] was last
    // read by the convert of batch b-2 (ev_q[s]); qf[s] by its kernels (ev_run[s]).
    if (b == 0 && C > 0) {
        need_pin(c, 0);
        SDPA_TRY(stage_chunk(pl, rk, rp, g, c.K, c.V, 0));
    }
    need_pin(c, 1);
    if (b >= 2) {
        HIP_TRY(hipStreamWaitEvent(rk.s_cp, rk.ev_q[s], 0));
        HIP_TRY(hipStreamWaitEvent(rk.s_in, rk.ev_run[s], 0));
        // host converts: the copy itself writes the operand image qf[s], which batch b-2's kernels read
        if (HI.cv) HIP_TRY(hipStreamWaitEvent(rk.s_cp, rk.ev_run[s], 0));
    }
    if (head_pieces) {
        for (int j = 0; j < pieces; ++j)
            SDPA_TRY(stage_q_rows(pl, rk, c.Q, s, i0, j * pr, std::min(pr, bs - j * pr), rk.ev_qh[j], rk.ev_qp[j],
                                  HI.cv ? HI.q_task[g][b][j] : -1));
        HIP_TRY(hipEventRecord(rk.ev_q[s], rk.s_in));
    } else {
        SDPA_TRY(stage_q_rows(pl, rk, c.Q, s, i0, 0, bs, rk.ev_qh[0], rk.ev_q[s], HI.cv ? HI.q_task[g][b][0] : -1));
    }
    // the chunks behind the first: enqueued AFTER chunk 0's launches (see below), so that a rank's first
    // kernel is issued before the host spends time on the K/V remainders' registration
    // With HOST converts staging a chunk WAITS (on this thread) until the pool has written its rows: staging every
    // chunk before chunk 1's launch held that launch back until the whole shard was converted -- the compute stream
    // sat idle for 2.7 ms of config 3's call and 0.7 ms of the metric shape's (profiles/r04/boundary_timeline_*.txt).
    // So they are staged a few chunks ahead of the launch that needs them (device converts: all at once, nothing waits).
    bool rest_staged = b != 0;
    int staged_hi = 0;
    auto stage_upto = [&](int hi) -> int {
        if (rest_staged) return SDPA_OK;
        need_pin(c, 2);
        hi = std::min(hi, C - 1);
        for (int ch = staged_hi + 1; ch <= hi; ++ch) SDPA_TRY(stage_chunk(pl, rk, rp, g, c.K, c.V, ch));
        staged_hi = std::max(staged_hi, hi);
        if (staged_hi >= C - 1) {
            rest_staged = true;
            if (g == 0) HIP_TRY(hipEventRecord(rk.ev_kv_done, rk.s_in));
        }
        return SDPA_OK;
    };
    auto stage_rest = [&]() -> int { return stage_upto(C - 1); };

    // compute stream.  contrib[s] / stat[s] / out64[s] were last used by batch b-2: by its finish + D2H
    // (ev_out[s]) when this rank finishes its rows itself, by its collective tail on the comm stream
    // (ev_comm[s], recorded by the calling thread: wait until that tail has been ENQUEUED) otherwise.
    if (b >= 2) {
        if (finisher) {
            HIP_TRY(hipStreamWaitEvent(rk.s_run, rk.ev_out[s], 0));
        } else {
            while (c.tails.load(std::memory_order_acquire) < b - 1) {
                if (c.failed.load(std::memory_order_relaxed)) return SDPA_EHIP;
                cpu_relax();
            }
            HIP_TRY(hipStreamWaitEvent(rk.s_run, rk.ev_comm[s], 0));
        }
    }
    const bool streamed = b == 0 && rp.n_slots > 1;
    bool have_all_q = false;
    auto need_all_q = [&]() -> int {
        if (!have_all_q) HIP_TRY(hipStreamWaitEvent(rk.s_run, rk.ev_q[s], 0));
        have_all_q = true;
        return SDPA_OK;
    };
    const int n_launch_chunks = streamed ? C : 1;
    for (int ch = 0; ch < n_launch_chunks; ++ch) {
        const bool first = ch == 0, last = ch + 1 == n_launch_chunks;
        const bool in_pieces = pieces > 1 && ((first && head_pieces) || (last && tail_pieces));
        const int k0 = streamed ? rp.chunks[ch].k0 : 0;
        const int keys = streamed ? rp.chunks[ch].keys : rp.key_cnt;
        // a launch that needs chunks behind the first (a later chunk, or the whole shard at once)
        if (!(streamed && first)) SDPA_TRY(stage_upto((streamed && HI.cv) ? ch + kStageAhead : C - 1));
        if (b == 0 && C > 0) HIP_TRY(hipStreamWaitEvent(rk.s_run, rk.ev_kv[streamed ? ch : C - 1], 0));
        const int np = in_pieces ? pieces : 1;
        for (int j = 0; j < np; ++j) {
            const int j0 = in_pieces ? j * pr : 0, jr = in_pieces ? std::min(pr, bs - j0) : bs;
            if (first && head_pieces && in_pieces)
                HIP_TRY(hipStreamWaitEvent(rk.s_run, rk.ev_qp[j], 0));
            else
                SDPA_TRY(need_all_q());
            int sp, slot0 = -1;
            if (streamed) {
                sp = rp.chunks[ch].splits;                // the slot count was planned for this launch shape
                slot0 = rp.chunks[ch].slot0;
            } else {
                sp = keys > 0 ? pick_splits(pl, jr, keys) : 1;
            }
            SDPA_TRY(bracket());
            SDPA_TRY(launch_fused(pl, rk, rp, s, bs, j0, jr, k0, keys, sp, slot0));
            SDPA_TRY(bracket());
            if (c.first_kernel_us[g] == 0.0) c.first_kernel_us[g] = now_us() - c.t_enter;
            if (g == 0) {
                c.last_splits = sp;
                c.last_rows = jr;
                c.last_keys = keys;
                c.last_note = sdpa::last_launch_note();
            }
            if (last) {
                if (streamed) SDPA_TRY(merge_slots(pl, rk, rp, s, bs, j0, jr));
                if (finisher) SDPA_TRY(finish_rows(c, g, s, bs, i0, in_pieces ? j : 0, j0, jr));
            }
        }
    }
    SDPA_TRY(stage_rest());
    HIP_TRY(hipEventRecord(rk.ev_run[s], rk.s_run));
    if (finisher) HIP_TRY(hipEventRecord(rk.ev_out[s], rk.s_out));
    return SDPA_OK;
}

// K/V plan over P ranks: merge the shard-local triples of batch b (attention-mpi.c:340-380) on the
// ranks' COMM streams -- batch b+1's fused kernels are already running on the compute streams, the
// way the reference leaves its MPI_Ireduce in flight under the next batch (:364-380).
int tail_batch(Call &c, int b) {
    const Plan &pl = c.pl;
    const int P = pl.P, s = b & 1, dv = c.dv;
    Rank &root = E.r[0];
    const int bs = std::min(pl.B, c.m - b * pl.B);
    const size_t i0 = (size_t)b * pl.B;
    std::vector<float *> send(P), recv(P);
    std::vector<hipStream_t> comm(P);
    for (int g = 0; g < P; ++g) {
        Rank &rk = E.r[g];
        comm[g] = rk.s_comm;
        HIP_TRY(hipSetDevice(rk.dev));
        HIP_TRY(hipStreamWaitEvent(rk.s_comm, rk.ev_run[s], 0));     // the rank's partial triple of batch b
        if (b >= 2) HIP_TRY(hipStreamWaitEvent(rk.s_comm, rk.ev_out[s], 0));   // out64[s] of batch b-2 still leaving
    }
    // rank 0, last batch: where the tail's time goes (sdpa_timing.merge_us / reduce_us / egress_us)
    const bool mark = b == pl.nb - 1;
    auto mark_tail = [&](int i) -> int {
        if (!mark) return SDPA_OK;
        HIP_TRY(hipSetDevice(root.dev));
        HIP_TRY(hipEventRecord(root.ev_tail[i], root.s_comm));
        return SDPA_OK;
    };
    SDPA_TRY(mark_tail(0));
    if (pl.merge_allreduce) {
        // :342 gmax = allreduce MAX(lmax); :346-351 rescale; :354 gsum = allreduce SUM(lsum);
        // :358-362 normalise
        for (int g = 0; g < P; ++g) {
            send[g] = (float *)E.r[g].stat[s].p;
            recv[g] = (float *)E.r[g].gstat[s].p;
        }
        if (E.coll->all_reduce(send.data(), recv.data(), bs, RedOp::Max, comm.data())) return coll_fail();
        for (int g = 0; g < P; ++g) {
            Rank &rk = E.r[g];
            HIP_TRY(hipSetDevice(rk.dev));
            HIP_TRY(sdpa::launch_merge_rescale((float *)rk.contrib[s].p, pl.ldo, (float *)rk.stat[s].p + bs,
                                               (const float *)rk.stat[s].p, (const float *)rk.gstat[s].p, bs,
                                               dv, rk.s_comm));
            send[g] = (float *)rk.stat[s].p + bs;
            recv[g] = (float *)rk.gstat[s].p + bs;
        }
        if (E.coll->all_reduce(send.data(), recv.data(), bs, RedOp::Sum, comm.data())) return coll_fail();
        for (int g = 0; g < P; ++g) {
            Rank &rk = E.r[g];
            HIP_TRY(hipSetDevice(rk.dev));
            HIP_TRY(sdpa::launch_merge_normalise((float *)rk.contrib[s].p, pl.ldo,
                                                 (const float *)rk.gstat[s].p + bs, bs, dv, rk.s_comm));
        }
    } else {
        // one all-gather of the (lmax, lsum) pairs, then steps 2-5 in one pass on every rank
        for (int g = 0; g < P; ++g) {
            send[g] = (float *)E.r[g].stat[s].p;
            recv[g] = (float *)E.r[g].gstat[s].p;
        }
        if (E.coll->all_gather(send.data(), recv.data(), 2 * (size_t)bs, comm.data())) return coll_fail();
        for (int g = 0; g < P; ++g) {
            Rank &rk = E.r[g];
            HIP_TRY(hipSetDevice(rk.dev));
            HIP_TRY(sdpa::launch_merge_gathered((float *)rk.contrib[s].p, pl.ldo, (const float *)rk.gstat[s].p,
                                                P, g, bs, dv, rk.s_comm));
        }
    }
    SDPA_TRY(mark_tail(1));
    need_pin(c, 3);
    for (int g = 0; g < P; ++g) send[g] = (float *)E.r[g].contrib[s].p;
    c.tail_marked = mark;
    if (pl.egress_scatter) {
        // sum of the normalised contributions, SCATTERED: rank r receives rows [r*share, (r+1)*share) of the
        // batch, widens them (:373/:396) and sends them home over its own PCIe link -- P links instead of
        // the root's one (the reference funnels to rank 0 because an MPI rank has no other way, :379-380)
        const int share = (bs + P - 1) / P;
        for (int g = 0; g < P; ++g) recv[g] = (float *)E.r[g].red[s].p;
        if (E.coll->reduce_scatter_sum(send.data(), recv.data(), (size_t)share * pl.ldo, comm.data())) return coll_fail();
        SDPA_TRY(mark_tail(2));
        for (int g = 0; g < P; ++g) {
            Rank &rk = E.r[g];
            const int r0 = g * share, rows = std::min(share, bs - r0);
            HIP_TRY(hipSetDevice(rk.dev));
            if (c.widen) {
                // (the repack, if the rows are padded, runs on the comm stream; ev_comm[s] doubles as "produced")
                if (g == 0 && mark) HIP_TRY(hipEventRecord(root.ev_tail[3], root.s_comm));
                SDPA_TRY(ship_rows_f32(c, rk, i0 + r0, rows, (const float *)rk.red[s].p, pl.ldo, nullptr,
                                       (float *)rk.out64[s].p, rk.s_comm, rk.ev_comm[s], rk.s_out));
                if (rows <= 0) {
                    HIP_TRY(hipEventRecord(rk.ev_comm[s], rk.s_comm));
                    HIP_TRY(hipStreamWaitEvent(rk.s_out, rk.ev_comm[s], 0));
                }
                HIP_TRY(hipEventRecord(rk.ev_out[s], rk.s_out));
                continue;
            }
            if (rows > 0)
                HIP_TRY(sdpa::launch_cvt_f2d((const float *)rk.red[s].p, pl.ldo, (double *)rk.out64[s].p, rows, dv,
                                             rk.s_comm));
            if (g == 0 && mark) HIP_TRY(hipEventRecord(root.ev_tail[3], root.s_comm));
            HIP_TRY(hipEventRecord(rk.ev_comm[s], rk.s_comm));
            HIP_TRY(hipStreamWaitEvent(rk.s_out, rk.ev_comm[s], 0));
            if (rows > 0)
                SDPA_TRY(copy_result_rows(c, c.result + (i0 + r0) * dv, rk.out64[s].p, (size_t)rows * dv * sizeof(double),
                                          rk.s_out));
            HIP_TRY(hipEventRecord(rk.ev_out[s], rk.s_out));
        }
    } else {
        // :380 reduce(SUM) of the normalised contributions to rank 0, :373/:396 widen, D2H
        if (E.coll->reduce_sum_to_root(send.data(), (float *)root.red[s].p, (size_t)bs * pl.ldo, comm.data()))
            return coll_fail();
        SDPA_TRY(mark_tail(2));
        HIP_TRY(hipSetDevice(root.dev));
        if (!c.widen)
            HIP_TRY(sdpa::launch_cvt_f2d((const float *)root.red[s].p, pl.ldo, (double *)root.out64[s].p, bs, dv,
                                         root.s_comm));
        SDPA_TRY(mark_tail(3));
        for (int g = 0; g < P; ++g) {
            if (g == 0 && c.widen) continue;                                // (the root's are recorded by ship_rows_f32)
            HIP_TRY(hipSetDevice(E.r[g].dev));
            HIP_TRY(hipEventRecord(E.r[g].ev_comm[s], E.r[g].s_comm));
            HIP_TRY(hipEventRecord(E.r[g].ev_out[s], E.r[g].s_comm));      // (non-roots: nothing leaves)
        }
        HIP_TRY(hipSetDevice(root.dev));
        if (c.widen) {
            SDPA_TRY(ship_rows_f32(c, root, i0, bs, (const float *)root.red[s].p, pl.ldo, nullptr, (float *)root.out64[s].p,
                                   root.s_comm, root.ev_comm[s], root.s_out));
        } else {
            HIP_TRY(hipStreamWaitEvent(root.s_out, root.ev_comm[s], 0));
            SDPA_TRY(copy_result_rows(c, c.result + i0 * dv, root.out64[s].p, (size_t)bs * dv * sizeof(double), root.s_out));
        }
        HIP_TRY(hipEventRecord(root.ev_out[s], root.s_out));
    }
    return SDPA_OK;
}

// one rank's enqueue thread: batch after batch, at most one batch ahead of the collective tails
int rank_thread(void *arg, int g) {
    Call &c = *(Call *)arg;
    for (int b = 0; b < c.pl.nb; ++b) {
        if (c.failed.load(std::memory_order_relaxed)) break;
        const int rc = rank_batch(c, g, b);
        if (rc != SDPA_OK) {
            c.fail(rc);
            break;
        }
        c.enq[g].store(b + 1, std::memory_order_release);
    }
    return SDPA_OK;
}

// ---- where the fp64 -> operand converts run ($SDPA_HOST_CVT) ----------------------------------------
// 0 = always on the device; 1 = always on host threads; unset / "auto" = per problem: on the host when the
// call would otherwise wait for PCIe -- the fp64 inputs take clearly longer over the link than the
// kernels take (BASELINE config 5 in bf16: 671 MB in for 3.6 ms of kernel, boundary 15.2 -> 9.1 ms;
// config 2: 1.15 -> 0.92 ms) -- and on the device when the kernels cover the transfer anyway (the
// metric shape: 9.6 vs 10.0-10.9 ms) or several ranks share the host's convert threads
// (profiles/r03/host_convert_ab.log).
int host_cvt_mode() {
    const char *v = getenv("SDPA_HOST_CVT");
    if (!v || !*v || strcmp(v, "auto") == 0) return 2;
    return atoi(v) > 0 ? 1 : 0;
}

// CPUs this process can really keep busy: its affinity mask cut down to the container's CPU quota (cgroup v2 cpu.max, v1
// cpu.cfs_quota_us / cpu.cfs_period_us).  std::thread::hardware_concurrency() says 256 on a GPU box whose container may use
// 16 cores' worth of CPU time.  $SDPA_DEBUG host_cores overrides (tests pin the model with it).
int effective_cores() {
    if (sdpa_debug_int("host_cores", 0) > 0) return sdpa_debug_int("host_cores", 0);
    static const int cores = [] {
        int n = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = CPU_COUNT(&set);
        double quota = 0.0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            long period = 0;
            if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) quota = atof(q) / (double)period;
            fclose(f);
        } else {
            long q = -1, period = 0;
            if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%ld", &q) != 1) q = -1; fclose(fq); }
            if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%ld", &period) != 1) period = 0; fclose(fp); }
            if (q > 0 && period > 0) quota = (double)q / (double)period;
        }
        if (quota >= 1.0 && quota < n) n = (int)(quota + 0.5);
        return n < 1 ? 1 : n;
    }();
    return cores;
}

// two threads per usable core (the rows are half load, half convert/store: 32 threads on a 16-core quota measured ~100 GB/s of
// fp64 source), at most 128
int host_convert_thread_count() {
    const int cores = effective_cores();
    const int dflt = cores > 4 ? std::min(128, 2 * cores) : std::max(1, cores - 1);
    return env_int("SDPA_HOST_CVT_THREADS", dflt);
}

// ---- the feed model (round 6): where the fp64 -> operand converts of a call should run, for P ranks -----------
// ONE converter pool serves every rank of the process, each rank has its OWN PCIe link and its own kernels:
//   t_host   = every rank's inputs (K, V once; Q once -- the K/V plan's ranks read the same rows) / the pool's rate
//   t_link   = ONE rank's fp64 inputs (its K/V shard + every Q row it scores) / 55 GB/s -- what device converts pull over the link
//   t_kernel = one rank's fused kernels over the whole call
// Host converts (and with them the streamed first batch) pay while the pool is not what everybody waits for.  At P = 1 the rules
// of rounds 3-5 stand.  At P > 1 page-locked arrays go to the device converts + the launch-per-chunk schedule once
// t_host > 1.1 max(t_kernel, t_link).  Measured (profiles/r06/feed_model_p8.log): on the GPU boxes' host the pool is NOT what P = 8
// waits for at the BASELINE shapes -- config 3: 570 MB in 2.9-3.6 ms against 4.2 ms of kernel per rank; the metric shape: 0.9-1.1 ms
// against 1.06 ms of kernel and 0.92 ms of link -- so they keep host converts and the streamed launches; a host with a slower pool
// (fewer usable cores) or a shape with less kernel per byte flips to the device converts.  Pageable arrays keep the pool whatever P is (a pageable source makes the
// runtime stage through ITS bounce buffers on the enqueueing thread: slower than the pool at any P) -- the pool itself is sized
// by the cores the process may really use.  sdpa_plan_describe() prints the three times and both decisions.
struct FeedModel {
    int cores = 0, threads = 0;
    double pool_Bps = 0.0, t_host = 0.0, t_link = 0.0, t_kernel = 0.0;
    bool host_ok = false, streamable = false;
};
FeedModel feed_model(const Plan &pl) {
    FeedModel f;
    f.cores = effective_cores();
    f.threads = host_convert_thread_count();
    f.host_ok = f.cores >= 16 && f.threads >= 8;
    // 5.0-6.4 GB/s of fp64 source per busy thread, measured with 8 loopback ranks' worth of work on the pool (config 3: 570 MB in
    // 2.9-3.6 ms = 160-194 GB/s over the span with 32 threads; metric shape 168 MB in 0.90-1.09 ms; config 4 268 MB in 1.34-1.74 ms:
    // profiles/r06/feed_model_p8.log), up to what the host's memory delivers to streaming readers.  (The boxes' 16-core cgroup quota
    // is CPU TIME per 100 ms period -- 1.6 CPU-seconds: a 3 ms burst of 32 threads uses 0.1 of it and is not throttled; the quota
    // sizes the pool, it does not slow a call down.)
    f.pool_Bps = std::min(240e9, std::max(1, f.threads) * 5.3e9);
    const double P = std::max(1, pl.P);
    const double m_rank = pl.qrows ? (double)pl.m / P : (double)pl.m;
    const double keys_rank = pl.qrows ? (double)pl.n : (double)pl.n / P;
    f.t_host = ((double)pl.n * (pl.dk + pl.dv) + (double)pl.m * pl.dk) * 8.0 / f.pool_Bps;
    f.t_link = (keys_rank * (pl.dk + pl.dv) + m_rank * pl.dk) * 8.0 / 55e9;           // fp64 over PCIe Gen5 x16, as measured
    const double rate = pl.bf16 ? 1.0e15 : (pl.dk <= 256 ? 1.3e14 : 1.0e14);
    f.t_kernel = 2.0 * m_rank * keys_rank * (pl.dk + pl.dv) / rate;
    for (const RankPlan &rp : pl.r) f.streamable = f.streamable || rp.stream.on;
    return f;
}

// `pageable`: the caller's input arrays are neither page-locked already nor about to be registered -- the device converts
// would pull fp64 from pageable memory through the runtime's staging copies (metric shape 11.8 ms, config 3 44.3 ms at
// the boundary, against 9.7 / 37.0 with host converts and 9.8 / 34.5 registered: profiles/r04/host_register_ab.log)
bool want_host_cvt(const Plan &pl, bool pageable) {
    const int mode = host_cvt_mode();
    if (mode != 2) return mode == 1;
    const double elems = (double)pl.m * pl.dk + (double)pl.n * pl.dk + (double)pl.n * pl.dv;
    if (elems < 1e6) return false;                                   // latency bound either way
    const FeedModel f = feed_model(pl);
    if (pageable) return f.host_ok;
    if (pl.P == 1) {
        // page-locked caller arrays: the streamed first batch (StreamPlan) can only be fed by the copy engine -- a device
        // convert kernel could not run beside the persistent launch -- and the host converts cost the call nothing where
        // the kernels cover the transfer (metric shape 9.0 vs 9.1-9.2 ms, profiles/r04/hostlevel_all_configs.log)
        if (f.streamable && f.host_ok) return true;
        // ... otherwise only where the HOST can do it faster than the link would have carried the fp64 bytes
        if (!f.host_ok) return false;
        return f.t_link > 1.4 * f.t_kernel && f.t_host < f.t_link;
    }
    // P > 1, page-locked: the streamed launches with host converts while ONE pool keeps up with P ranks, else every rank pulls
    // its fp64 shard over its own link and converts on the device (launch per chunk)
    if (!f.streamable || !f.host_ok) return false;
    return f.t_host <= 1.1 * std::max(f.t_kernel, f.t_link);
}

// ---- where the result is widened to fp64 ($SDPA_HOST_WIDEN) ---------------------------------------------
// 0 = on the device (finish_f64 / cvt_f2d kernels, fp64 rows cross PCIe into the registered `result`); 1 = on host
// threads: fp32 rows cross PCIe into page-locked staging and the converter pool widens them into `result`, the
// reference's own placement (cvt_f2d_avx512 on the root, attention-mpi.c:373 / :396) -- half the D2H bytes, and
// `result` is never registered (registering a never-touched array faults all its pages in on the calling thread);
// unset / "auto": on the host when the host has the threads for it (>= 16 hardware threads, >= 8 in the pool) and
// the result is big enough to be worth waking them (profiles/r04/host_widen_ab.log).
int host_widen_mode() {
    const char *v = getenv("SDPA_HOST_WIDEN");
    if (!v || !*v || strcmp(v, "auto") == 0) return 2;
    return atoi(v) > 0 ? 1 : 0;
}

// `pageable_result`: the caller's result array is neither page-locked nor about to be registered.  A device-to-host
// copy straight into it would BLOCK the enqueuing thread until the rows exist (the runtime stages pageable
// destinations synchronously), serialising batch b+1's enqueue behind batch b's kernels: such a result
// comes home through the page-locked staging whatever the host's thread count -- a small host widens with the few
// threads it has, on the calling thread if need be.
bool want_host_widen(const Plan &pl, bool pageable_result = true) {
    const int mode = host_widen_mode();
    if (mode != 2) return mode == 1;
    if ((double)pl.m * pl.dv < 256.0 * 1024.0) return false;
    if (pageable_result) return true;
    return effective_cores() >= 16 && host_convert_thread_count() >= 8;
}

// the converter pool is created on first use (32 threads, $SDPA_HOST_CVT_THREADS)
int ensure_host_converter() {
    if (E.hc) return SDPA_OK;
    E.hc = sdpa::HostConverter::create(host_convert_thread_count());
    return E.hc ? SDPA_OK : SDPA_ENOMEM;
}

// The lazy default (no sdpa_init, no $SDPA_GPUS): EVERY visible device, as the reference uses every rank it is
// given (attention-mpi.c:199) -- on the evidence of this very node: creating the engine on P > 1 devices runs
// every collective of the pipeline once over RCCL on known data (sdpa_coll.hip: selftest).  Passed: P GPUs.
// Failed with an error: one line on stderr and ONE GPU (the round-3 default).  Did not finish (a transport
// that hangs): SDPA_ERCCL with the advice to set SDPA_GPUS=1 -- the devices may still hold its kernels.
int lazy_init() {
    if (E.up) return SDPA_OK;
    if (const char *sdma = getenv("HSA_ENABLE_SDMA"))
        if (*sdma && atoi(sdma) == 0) E.stream_off = true;      // no copy engines: every copy is a shader, none can run beside a resident launch
    if (const char *env = getenv("SDPA_GPUS")) {
        const int want = (strcmp(env, "all") == 0 || strcmp(env, "0") == 0) ? 0 : atoi(env);
        return sdpa_init(want < 0 ? 1 : want);
    }
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) (void)hipGetLastError();
    if (cnt <= 1 || getenv("SDPA_VIRTUAL_GPUS")) return sdpa_init(1);
    const int rc = sdpa_init(cnt > sdpa::kMaxRanks ? sdpa::kMaxRanks : cnt);
    if (rc == SDPA_OK) return rc;
    if (E.rccl_hung) {
        fprintf(stderr, "sdpa: the RCCL self-test over %d GPUs did not finish; set SDPA_GPUS=1 to run on one GPU\n", cnt);
        return rc;
    }
    fprintf(stderr, "sdpa: engine on all %d visible GPUs failed (%s); using ONE GPU (SDPA_GPUS=N forces a count)\n", cnt,
            sdpa_strerror(rc));
    return sdpa_init(1);
}

int check_shape(const void *Q, const void *K, const void *V, const void *result, int m, int n, int dk,
                int dv, int flags, bool need_ptrs) {
    if (need_ptrs && (!Q || !K || !V || !result)) return SDPA_EINVAL;
    if (m <= 0 || n <= 0 || dk <= 0 || dv <= 0) return SDPA_EINVAL;
    (void)flags;           // no shape is refused any more: bf16 beyond its kernels' dims runs fp32 (bf16_for), any dk has a kernel
    return SDPA_OK;
}

// What only the plan knows: the bf16 kernels carry Vt byte offsets in 32 bits and a rank's Vt image
// spans its whole shard (ldvt = pad_n(key_cnt)), so padded dv x shard keys x 2 bytes must stay below
// 4 GiB (dv = 1024: ~2 M keys per rank).  Refused HERE, before anything is pinned, copied or queued --
// the launcher's own check would fire in the middle of the pipeline (sdpa_dev_shard_partial_bf16
// refuses the same shapes at the device level).
int check_plan(const Plan &pl) {
    if (!pl.bf16) return SDPA_OK;
    for (const RankPlan &rp : pl.r)
        if ((double)sdpa::bf16_pad_dv(pl.dv) * (double)sdpa::bf16_pad_n(rp.key_cnt) * 2.0 >= 4294967296.0) {
            fprintf(stderr, "sdpa: bf16 path: dv=%d with %d keys on one rank exceeds the kernels' 32-bit Vt offsets\n",
                    pl.dv, rp.key_cnt);
            return SDPA_EUNSUP;
        }
    return SDPA_OK;
}

void destroy_rank(Rank &g) {
    if (hipSetDevice(g.dev) != hipSuccess) return;
    (void)hipDeviceSynchronize();
    if (g.vf_view) { g.vf = DevBuf(); g.vf_view = false; }
    DevBuf *single[] = {&g.k64, &g.v64, &g.kf, &g.vf, &g.ws, &g.slots};
    for (DevBuf *b : single) if (b->p) (void)hipFree(b->p);
    for (int s = 0; s < 2; ++s) {
        DevBuf *pair[] = {&g.q64[s], &g.qf[s], &g.contrib[s], &g.stat[s], &g.gstat[s], &g.red[s], &g.out64[s]};
        for (DevBuf *b : pair) if (b->p) (void)hipFree(b->p);
        hipEvent_t evs[] = {g.ev_q[s], g.ev_run[s], g.ev_out[s], g.ev_comm[s]};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : g.ev_sub[s]) if (e) (void)hipEventDestroy(e);
    }
    for (hipEvent_t e : g.ev_kv) (void)hipEventDestroy(e);
    for (hipEvent_t e : g.ev_h2d) (void)hipEventDestroy(e);
    for (hipEvent_t e : g.ev_qh) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : g.ev_qp) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : g.ev_k) (void)hipEventDestroy(e);
    for (hipEvent_t e : g.ev_w) (void)hipEventDestroy(e);
    if (g.sflags) (void)hipFree(g.sflags);
    if (g.h_gen) (void)hipHostFree(g.h_gen);
    if (g.h_status) (void)hipHostFree(g.h_status);
    hipEvent_t evs[] = {g.ev_t0, g.ev_kv_done, g.ev_end, g.ev_tail[0], g.ev_tail[1], g.ev_tail[2], g.ev_tail[3]};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    if (g.s_cp) (void)hipStreamDestroy(g.s_cp);
    if (g.s_in) (void)hipStreamDestroy(g.s_in);
    if (g.s_run) {
        sdpa::forget_stream_cus(g.s_run);
        (void)hipStreamDestroy(g.s_run);
    }
    if (g.s_out) (void)hipStreamDestroy(g.s_out);
    if (g.s_comm) (void)hipStreamDestroy(g.s_comm);
    g = Rank();
}

int create_rank(Rank &g, int dev, int reserve) {
    g.dev = dev;
    HIP_TRY(hipSetDevice(dev));
    // every kernel's code object on this device now (sdpa_internal.h: preload_kernels_*)
    HIP_TRY(sdpa::preload_kernels_f32());
    HIP_TRY(sdpa::preload_kernels_dksplit());
    HIP_TRY(sdpa::preload_kernels_bf16());
    HIP_TRY(sdpa::preload_kernels_aux());
    HIP_TRY(sdpa::preload_kernels_coll());
    // converts go first when a fused launch retires: they feed the next one
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(hipStreamCreateWithFlags(&g.s_cp, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithPriority(&g.s_in, hipStreamNonBlocking, hi));
    // $SDPA_COMM_CUS=R (opt-in): the compute stream leaves R compute units (rounded up to whole
    // multiples of the 8 XCDs) to everything else.  A fused launch otherwise holds every wave slot of
    // every CU until it ends (two 64 KiB workgroups and the whole register file per CU), so a merge
    // collective, an RCCL kernel or a convert that becomes ready while it runs waits for its last
    // workgroup -- streams and priorities do not help a kernel that finds no CU (profiles/r03/).
    // Round 4: the fused kernels distribute their work by stream-K over whatever CUs the stream has, so the
    // reservation no longer breaks the grid's fit -- it is ON (8 CUs) by default for engines of several ranks.
    if (reserve > 0) {
        SDPA_TRY(sdpa::create_masked_stream(&g.s_run, reserve));
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&g.s_run, hipStreamNonBlocking));
    }
    HIP_TRY(hipStreamCreateWithPriority(&g.s_out, hipStreamNonBlocking, hi));
    HIP_TRY(hipStreamCreateWithPriority(&g.s_comm, hipStreamNonBlocking, hi));
    for (int s = 0; s < 2; ++s) {
        HIP_TRY(hipEventCreateWithFlags(&g.ev_comm[s], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g.ev_q[s], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g.ev_run[s], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g.ev_out[s], hipEventDisableTiming));
        for (int j = 0; j < kMaxSub; ++j) HIP_TRY(hipEventCreateWithFlags(&g.ev_sub[s][j], hipEventDisableTiming));
    }
    for (int j = 0; j < kMaxSub; ++j) {
        HIP_TRY(hipEventCreateWithFlags(&g.ev_qh[j], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g.ev_qp[j], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreate(&g.ev_t0));
    HIP_TRY(hipEventCreate(&g.ev_kv_done));
    HIP_TRY(hipEventCreate(&g.ev_end));
    for (hipEvent_t &e : g.ev_tail) HIP_TRY(hipEventCreate(&e));
    return SDPA_OK;
}

int init_impl(int n_gpus) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
        (void)hipGetLastError();
        fprintf(stderr, "sdpa: no HIP device visible\n");
        return SDPA_ENODEV;
    }
    const int virt = env_int("SDPA_VIRTUAL_GPUS", 0);
    if (virt > sdpa::kMaxRanks) {
        fprintf(stderr, "sdpa: SDPA_VIRTUAL_GPUS=%d exceeds %d\n", virt, sdpa::kMaxRanks);
        return SDPA_EINVAL;
    }
    const int want = virt > 0 ? virt : (n_gpus == 0 ? cnt : n_gpus);
    if (virt == 0 && want > cnt) {
        fprintf(stderr, "sdpa: %d GPUs requested, %d visible\n", want, cnt);
        return SDPA_ENODEV;
    }
    if (want > sdpa::kMaxRanks) return SDPA_EINVAL;
    if (E.up && E.n == want && E.virtual_ranks == (virt > 0)) return SDPA_OK;
    if (E.up) sdpa_shutdown();

    E.r.assign(want, Rank());
    E.virtual_ranks = virt > 0;
    for (int i = 0; i < want; ++i) {
        const int dev = virt > 0 ? 0 : i;
        HIP_TRY(hipSetDevice(dev));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            fprintf(stderr, "sdpa: device %d is %s; this engine is built for gfx950 only\n", dev,
                    prop.gcnArchName);
            return SDPA_ENODEV;
        }
        SDPA_TRY(create_rank(E.r[i], dev, comm_cus_reserved(prop.multiProcessorCount, want)));
        if (i == 0) {
            E.run_cus = sdpa::stream_cus(E.r[0].s_run);
            E.chip_cus = prop.multiProcessorCount;
        }
    }
    const bool force = sdpa_debug_int("force_collectives", 0) != 0;
    if (want > 1 || force) {
        if (virt > 0) {
            E.coll = sdpa::make_loopback_collectives(want, 0);
        } else {
            std::vector<int> devs(want);
            for (int i = 0; i < want; ++i) devs[i] = i;
            E.coll = sdpa::make_rccl_collectives(want, devs.data(), &E.rccl_hung);
        }
        if (!E.coll) return SDPA_ERCCL;
    }
    if (want > 1 && sdpa_debug_int("enqueue_threads", 1) != 0) {
        std::vector<int> devs(want);
        for (int i = 0; i < want; ++i) devs[i] = E.r[i].dev;
        E.pool.start(want, devs);
    }
    if (!E.bounce && hipHostMalloc((void **)&E.bounce, 2 * 4096, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        E.bounce = nullptr;              // (the result's edge pages then travel as pageable copies)
    }
    E.n = want;
    E.up = true;
    return SDPA_OK;
}

}  // namespace

// =============================================================================
// lifecycle
// =============================================================================
extern "C" {

void sdpa_shutdown(void) {
    DeviceRestore restore;
    PF.reset();
    E.pool.shutdown();
    delete E.hc;                         // (its page-locked staging goes with it: before the ranks' devices are released)
    E.hc = nullptr;
    for (Rank &g : E.r) destroy_rank(g);
    E.r.clear();
    delete E.coll;
    E.coll = nullptr;
    if (E.bounce) (void)hipHostFree(E.bounce);
    E.bounce = nullptr;
    E.n = 0;
    E.run_cus = 0;
    E.chip_cus = 0;
    E.up = false;
    E.virtual_ranks = false;
    E.stream_off = false;                // (a new engine probes again)
    E.stream_timeout_ms_once = 0;
}

int sdpa_init(int n_gpus) {
    if (n_gpus < 0) return SDPA_EINVAL;
    sdpa::reload_launch_knobs();
    DeviceRestore restore;
    const int rc = init_impl(n_gpus);
    if (rc != SDPA_OK && !E.up) {        // a half-built engine is torn down, not leaked
        E.pool.shutdown();
        delete E.hc;
        E.hc = nullptr;
        for (Rank &g : E.r) destroy_rank(g);
        E.r.clear();
        delete E.coll;
        E.coll = nullptr;
        E.n = 0;
    }
    return rc;
}

int sdpa_init_default(void) {
    sdpa::reload_launch_knobs();
    DeviceRestore restore;
    return lazy_init();
}

int sdpa_engine_ranks(void) { return E.up ? E.n : 0; }

int sdpa_last_timing_sized(struct sdpa_timing *out, size_t size) {
    if (!out || size == 0) return SDPA_EINVAL;
    memcpy(out, &E.last, std::min(size, sizeof(sdpa_timing)));
    return SDPA_OK;
}

int sdpa_last_timing(struct sdpa_timing *out) { return sdpa_last_timing_sized(out, sizeof(sdpa_timing)); }

// =============================================================================
// host level
// =============================================================================
static int attention_call(const double *Q, const double *K, const double *V, double *result, int m, int n, int dk, int dv, int flags);

int sdpa_attention_f64(const double *Q, const double *K, const double *V, double *result, int m,
                       int n, int dk, int dv, int flags) {
    int rc = attention_call(Q, K, V, result, m, n, dk, dv, flags);
    E.stream_timeout_ms_once = 0;
    if (rc == kStreamTimedOut) {             // (every device is drained and the host converters are idle: attention_call's exits)
        E.stream_off = true;
        rc = attention_call(Q, K, V, result, m, n, dk, dv, flags);
        if (rc == kStreamTimedOut) rc = SDPA_EHIP;          // (cannot happen: no plan streams any more)
    }
    return rc;
}

static int attention_call(const double *Q, const double *K, const double *V, double *result, int m, int n, int dk, int dv, int flags) {
    SDPA_TRY(check_shape(Q, K, V, result, m, n, dk, dv, flags, true));
    const double t_enter = now_us();
    sdpa::reload_launch_knobs();          // on the calling thread, before any enqueue thread runs
    DeviceRestore restore;
    SDPA_TRY(lazy_init());

    Call c;
    c.Q = Q; c.K = K; c.V = V; c.result = result;
    c.m = m; c.n = n; c.dk = dk; c.dv = dv;
    c.t_enter = t_enter;
    Plan &pl = c.pl;
    make_plan(pl, m, n, dk, dv, flags);
    if (want_bf16(flags) && !pl.bf16)
        fprintf(stderr, "sdpa: bf16 path asked for dk=%d dv=%d n=%d on %d rank(s): beyond the bf16 kernels (dk <= 512, dv <= 1024, "
                "32-bit Vt offsets); this call runs the fp32 path\n", dk, dv, n, pl.P);
    SDPA_TRY(check_plan(pl));
    SDPA_TRY(ensure_buffers(pl));
    const int P = pl.P;
    // K/V rows a sdpa_kv_prefetch() of THIS problem already moved are not moved again; a prefetch
    // of anything else is void (the staging images are about to be overwritten)
    if (PF.active && !PF.matches(K, V, m, n, dk, dv, flags)) PF.reset();
    if (PF.active) {          // and only if it was planned the way this call is (the knobs are environment)
        bool same = (int)PF.k_done.size() == P;
        for (int g = 0; same && g < P; ++g) same = PF.k_done[g].size() == pl.r[g].chunks.size();
        if (!same) PF.reset();
    }
    struct ClearPrefetch { ~ClearPrefetch() { PF.reset(); } } clear_prefetch;   // one-shot, also on errors

    // Destruction order on every exit: first the ranks' enqueue threads are waited for (they use `c` and
    // the host images), then every device is drained (no queued copy may still reference the caller's
    // arrays or the converters' staging), then the host converters are waited for and forgotten, then
    // c.pins unregisters the caller's arrays.
    struct EndHostCvt {
        ~EndHostCvt() {
            if (HI.cv) HI.cv->finish();          // no thread reads the caller's arrays once we return
            HI = HostImages();
        }
    } end_hostcvt;
    DrainOnExit drain;
    struct JoinThreads {
        Call &c;
        bool running = false;
        ~JoinThreads() {
            if (!running) return;
            c.fail(SDPA_EHIP);               // (a no-op after a clean finish: everything is enqueued by then)
            E.pool.wait();
        }
    } join{c};
    struct ClearCuts { ~ClearCuts() { CUT = PinCuts(); } } clear_cuts;
    CUT = PinCuts();
    c.do_pin = register_caller_arrays();
    {   // ($SDPA_DEBUG host_probe=1: a verdict about a pointer holds for ONE call -- the caller may have freed and reused the address)
        std::lock_guard<std::mutex> lk(HR.mu);
        HR.probed.clear();
    }
    const bool pageable_in = !c.do_pin && !(page_locked(K) && page_locked(V) && page_locked(Q));
    c.k_bytes = (size_t)n * dk * sizeof(double);
    c.v_bytes = (size_t)n * dv * sizeof(double);
    const bool want_progressive = sdpa_debug_int("progressive_pin", 1) != 0;
    if (c.do_pin && !PF.active && want_progressive) c.progressive = plan_progressive_pins(c);
    // every caller array: the partial pages at its two ends are never registered (HostPins::add), so a copy that
    // touches them is split at the first / last page boundary inside the array
    auto edge_cuts = [](const void *base, size_t bytes, std::vector<const char *> &cuts) {
        const char *b0 = (const char *)base, *end = b0 + bytes;
        const char *a0 = HostPins::page_up(b0), *a1 = HostPins::page_down(end);
        if (a1 <= a0) return;
        if (a0 > b0) cuts.push_back(a0);
        if (a1 < end) cuts.push_back(a1);
        std::sort(cuts.begin(), cuts.end());
        cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    };
    auto set_edge_cuts = [&]() {
        if (!c.do_pin) return;
        edge_cuts(K, c.k_bytes, CUT.k);
        edge_cuts(V, c.v_bytes, CUT.v);
        edge_cuts(Q, (size_t)m * dk * sizeof(double), CUT.q);
    };
    set_edge_cuts();
    c.threaded = P > 1 && !E.pool.th.empty();

    // $SDPA_HOST_CVT=1: host threads write the operand images; submit every conversion now, in the order
    // the copies will ask for them (every rank's chunk 0, the first Q batch, the other chunks, the other
    // batches), and let the enqueue code wait for each piece right before it copies it
    for (Rank &rk : E.r) rk.ev_w_used = 0;
    if (want_host_widen(pl, !c.do_pin && !page_locked(result)) && ensure_host_converter() == SDPA_OK) {
        c.w_base = (float *)E.hc->staging(3, (size_t)m * dv * sizeof(float));
        c.widen = c.w_base != nullptr;
    }
    HI = HostImages();
    if (!PF.active && want_host_cvt(pl, pageable_in) && ensure_host_converter() == SDPA_OK) {
        const size_t kel = pl.kv_elem, qel = pl.q_elem;
        const int ldv_h = pl.bf16 ? dv : pl.ldv;
        const size_t vel = pl.bf16 ? sizeof(unsigned short) : sizeof(float);
        // (fp32, every rank streamed with interleaved groups, images equally wide: K and V share ONE staging, group by group -- HostImages::pair)
        bool pair = !pl.bf16 && pl.ldk == pl.ldv && sdpa_debug_int("stream_pair", 1) != 0;
        for (int g = 0; g < P && pair; ++g)
            pair = pl.r[g].stream.on && pl.r[g].stream.interleaved && E.r[g].vf_view &&
                   (char *)E.r[g].vf.p == (char *)E.r[g].kf.p + (size_t)pl.r[g].key_cnt * pl.ldk * sizeof(float);
        char *hk = (char *)E.hc->staging(0, (size_t)n * pl.ldk * kel * (pair ? 2 : 1));
        // bf16, first batch streamed: the staging holds the Vt IMAGES themselves (padded dv rows x padded keys per rank, packed
        // entry by entry: Chunk::img_off), not dense rows -- all ranks or none (a call never mixes the two V stagings)
        bool bf16_images = pl.bf16;
        for (int g = 0; g < P && bf16_images; ++g) bf16_images = pl.r[g].stream.on;
        size_t hv_bytes = pair ? 64 : (size_t)n * ldv_h * vel;
        std::vector<size_t> hv_rank_off(P, 0);
        if (bf16_images) {
            hv_bytes = 0;
            for (int g = 0; g < P; ++g) {
                hv_rank_off[g] = hv_bytes;
                hv_bytes += (size_t)sdpa::bf16_pad_dv(dv) * sdpa::bf16_pad_n(pl.r[g].key_cnt) * sizeof(unsigned short);
            }
        }
        char *hv = (char *)E.hc->staging(1, hv_bytes);
        char *hq = (char *)E.hc->staging(2, (size_t)m * pl.ldq * qel);
        if (hk && hv && hq) {
            c.hostcvt = true;
            c.progressive = false;
            HI.streamed.assign(P, 0);
            for (int g = 0; g < P; ++g)
                if (pl.r[g].stream.on && (!pl.bf16 || bf16_images)) HI.streamed[g] = 1, c.streamed = true;
            HI.v_rank_off = hv_rank_off;
            HI.pair = pair;
            if (c.streamed) {
                if (++E.stream_gen == 0) ++E.stream_gen;
                c.stream_gen = E.stream_gen;
                c.stream_timeout_ms = E.stream_timeout_ms_once > 0 ? E.stream_timeout_ms_once : env_int("SDPA_STREAM_TIMEOUT_MS", 500);
                c.stream_timeout_ticks = (unsigned long long)c.stream_timeout_ms * 100000ull;
                c.stream_drop_word = sdpa_debug_int("stream_drop_word", 0) - 1;
            }
            CUT = PinCuts();
            set_edge_cuts();                 // (only `result` is registered in this mode; K / V / Q travel from the staging images)
            HI.cv = E.hc;
            HI.k = hk; HI.v = hv; HI.q = hq;
            HI.ldv_host = ldv_h;
            E.hc->begin();
            const sdpa::CvtKind kind = pl.bf16 ? sdpa::kCvtBf16 : sdpa::kCvtF32;
            const double qmult = pl.bf16 ? (double)(1.44269504088896340736f * (1.0f / sqrtf((float)dk))) : 1.0;
            HI.k_task.assign(P, {});
            HI.v_task.assign(P, {});
            HI.q_task.assign(P, {});
            // (a streamed rank stages rp.stream.entries -- row ranges in arrival order -- instead of rp.chunks)
            auto list_of = [&](int g) -> const std::vector<Chunk> & { return HI.streamed[g] ? pl.r[g].stream.entries : pl.r[g].chunks; };
            auto submit_chunk = [&](int g, int ch) {
                const RankPlan &rp = pl.r[g];
                const Chunk &cc = list_of(g)[ch];
                const size_t row0 = (size_t)rp.key_off + cc.k0;
                // (bf16, dv > 256: rows of the TILED K image -- chunk swizzle by the row's place in its tile; chunks and entries
                //  start on tile boundaries of the rank's image, so the task's own row count gives it)
                if (pair) {            // [K rows of the group][V rows of the group] at row 2 * row0 of the shared staging
                    HI.k_task[g].push_back(E.hc->submit(K + row0 * dk, hk + 2 * row0 * pl.ldk * kel, cc.keys, dk, pl.ldk, kind, 1.0));
                    HI.v_task[g].push_back(E.hc->submit(V + row0 * dv, hk + (2 * row0 + cc.keys) * pl.ldk * kel, cc.keys, dv, pl.ldv, kind, 1.0));
                    return;
                }
                HI.k_task[g].push_back(E.hc->submit(K + row0 * dk, hk + row0 * pl.ldk * kel, cc.keys, dk, pl.ldk,
                                                     pl.bf16 && sdpa::bf16_tiled(dv) ? sdpa::kCvtBf16Swz : kind, 1.0));
                if (pl.bf16 && HI.streamed[g])        // the entry's columns of the Vt image, as a packed block of the staging
                    HI.v_task[g].push_back(E.hc->submit_t(V + row0 * dv, (unsigned short *)(hv + hv_rank_off[g]) + cc.img_off, cc.keys,
                                                           cc.keys_pad, dv, sdpa::bf16_pad_dv(dv), cc.keys_pad));
                else
                    HI.v_task[g].push_back(E.hc->submit(V + row0 * dv, hv + row0 * ldv_h * vel, cc.keys, dv,
                                                         ldv_h, kind, 1.0));
            };
            auto submit_q = [&](int g, int b) {
                const RankPlan &rp = pl.r[g];
                HI.q_task[g].emplace_back();
                const int j_lo = b * pl.B;
                if (j_lo >= rp.row_cnt) return;
                const int bs = std::min(pl.B, rp.row_cnt - j_lo);
                const bool last_batch = j_lo + pl.B >= rp.row_cnt;
                const int pr = (b == 0 || (last_batch && !pl.collectives)) ? piece_rows_of(pl, bs) : bs;
                const int np = b == 0 ? (bs + pr - 1) / pr : 1;           // Q arrives in pieces with batch 0 only
                const size_t i0 = (size_t)rp.row_off + j_lo;
                for (int j = 0; j < np; ++j) {
                    const int j0 = b == 0 ? j * pr : 0, jr = b == 0 ? std::min(pr, bs - j0) : bs;
                    HI.q_task[g][b].push_back(E.hc->submit(Q + (i0 + j0) * dk, hq + (i0 + j0) * pl.ldq * qel, jr, dk,
                                                            pl.ldq, kind, qmult));
                }
            };
            const int q_ranks = pl.qrows ? P : 1;                          // K/V plan: every rank reads the same Q rows
            // first what every rank's first kernel needs: chunk 0 (streamed: every entry of group 0), then Q batch 0
            std::vector<size_t> head(P, 0);
            for (int g = 0; g < P; ++g) {
                const std::vector<Chunk> &l = list_of(g);
                if (l.empty()) continue;
                head[g] = 1;
                while (HI.streamed[g] && head[g] < l.size() && l[head[g]].group == 0) ++head[g];
                for (size_t ch = 0; ch < head[g]; ++ch) submit_chunk(g, (int)ch);
            }
            for (int g = 0; g < q_ranks; ++g) submit_q(g, 0);
            size_t max_chunks = 0;
            for (int g = 0; g < P; ++g) max_chunks = std::max(max_chunks, list_of(g).size());
            for (size_t ch = 1; ch < max_chunks; ++ch)
                for (int g = 0; g < P; ++g)
                    if (ch >= head[g] && ch < list_of(g).size()) submit_chunk(g, (int)ch);
            for (int b = 1; b < pl.nb; ++b)
                for (int g = 0; g < q_ranks; ++g) submit_q(g, b);
            for (int g = q_ranks; g < P; ++g) HI.q_task[g] = HI.q_task[0];
            {
                const void *arrays[3] = {K, V, Q};
                const size_t bytes[3] = {c.k_bytes, c.v_bytes, (size_t)m * dk * sizeof(double)};
                E.hc->place_near(arrays, bytes, 3);
            }
            E.hc->kick();
        }
    }

    Rank &root = E.r[0];
    HIP_TRY(hipSetDevice(root.dev));
    if (!c.threaded) pin_stage(c, 0);
    const double t_reg1 = now_us();
    HIP_TRY(hipEventRecord(root.ev_t0, root.s_cp));

    if (c.threaded) {
        // the ranks' threads enqueue; this thread registers the caller's arrays stage by stage, then
        // enqueues each batch's collective tail as soon as every rank has enqueued the batch's kernels
        join.running = true;
        E.pool.run(rank_thread, &c);
        for (int st = 0; st <= 3; ++st) pin_stage(c, st);
        int rc = SDPA_OK;
        for (int b = 0; b < pl.nb && rc == SDPA_OK; ++b) {
            for (int g = 0; g < P; ++g)
                while (c.enq[g].load(std::memory_order_acquire) <= b && !c.failed.load(std::memory_order_relaxed)) cpu_relax();
            if (c.failed.load()) break;
            if (pl.collectives) rc = tail_batch(c, b);
            c.tails.store(b + 1, std::memory_order_release);
        }
        if (rc != SDPA_OK) c.fail(rc);
        E.pool.wait();
        join.running = false;
        if (c.failed.load()) return c.failed.load();
    } else {
        for (int b = 0; b < pl.nb; ++b) {
            for (int g = 0; g < P; ++g) SDPA_TRY(rank_batch(c, g, b));
            if (pl.collectives) SDPA_TRY(tail_batch(c, b));
            c.tails.store(b + 1, std::memory_order_release);
        }
    }
    const double t_enqueued = now_us();

    // ---- the one wait of the call ------------------------------------------------------------
    HIP_TRY(hipSetDevice(root.dev));
    HIP_TRY(hipStreamWaitEvent(root.s_out, root.ev_run[(pl.nb - 1) & 1], 0));
    HIP_TRY(hipEventRecord(root.ev_end, root.s_out));
    double t_landed = 0.0;                             // host clock when the last fp32 piece was in the staging area
    if (c.widen) {
        if (!c.w_pieces.empty()) {
            // (everything but the last piece: widened while the GPU is still producing the later ones)
            const Call::WidenPiece last = c.w_pieces.back();
            c.w_pieces.pop_back();
            SDPA_TRY(widen_landed_pieces(c));
            c.w_pieces.assign(1, last);
            while (hipEventQuery(last.landed) == hipErrorNotReady) cpu_relax();
            (void)hipGetLastError();
            t_landed = now_us();
            SDPA_TRY(widen_landed_pieces(c));
        }
    }
    for (int g = 0; g < P; ++g) {
        HIP_TRY(hipSetDevice(E.r[g].dev));
        HIP_TRY(hipStreamSynchronize(E.r[g].s_run));
        HIP_TRY(hipStreamSynchronize(E.r[g].s_comm));
        HIP_TRY(hipStreamSynchronize(E.r[g].s_out));
        HIP_TRY(hipStreamSynchronize(E.r[g].s_in));
        HIP_TRY(hipStreamSynchronize(E.r[g].s_cp));
    }
    for (const Call::Sliver &sv : c.slivers) memcpy(sv.dst, sv.src, sv.bytes);     // result's partial first / last page
    drain.armed = false;
    if (c.streamed)
        for (int g = 0; g < P; ++g)
            if (pl.r[g].stream.on && E.r[g].h_status && *E.r[g].h_status != 0) {
                fprintf(stderr, "sdpa: rank %d: the streamed launch waited more than %d ms for ready word %d (%s): on this runtime copies "
                        "do not land beside a launch that owns the chip.  The call is re-run on the launch-per-chunk schedule, which "
                        "this process keeps from now on (SDPA_STREAMED=0 selects it from the start)\n", g, c.stream_timeout_ms,
                        *E.r[g].h_status - 1, *E.r[g].h_status - 1 >= sdpa::kStreamMaxChunks ? "a Q row piece" : "a K/V group");
                return kStreamTimedOut;
            }
    const double t_exit = now_us();
    const double host_tail_us = c.widen && t_landed > 0.0 ? t_exit - t_landed : 0.0;   // the last piece's widening: exposed

    HIP_TRY(hipSetDevice(root.dev));
    const int n_brackets = c.n_brackets;
    double kernel_ms = 0.0;
    float ms = 0.f;
    for (int i = 0; i + 1 < n_brackets; i += 2) {
        HIP_TRY(hipEventElapsedTime(&ms, root.ev_k[i], root.ev_k[i + 1]));
        kernel_ms += ms;
    }
    sdpa_timing &T = E.last;
    T = sdpa_timing();
    T.total_us = t_exit - t_enter;
    T.register_us = c.pins.us;
    T.kernel_us = kernel_ms * 1e3;
    if (n_brackets >= 2) {
        const int rank0_chunks = (int)pl.r[0].chunks.size();
        if (rank0_chunks > 0 && pl.r[0].row_cnt > 0) {
            HIP_TRY(hipEventElapsedTime(&ms, root.ev_t0, root.ev_kv_done));
            T.kv_stage_us = ms * 1e3;
        }
        HIP_TRY(hipEventElapsedTime(&ms, root.ev_t0, root.ev_k[0]));
        T.head_us = (t_reg1 - t_enter) + ms * 1e3;
        HIP_TRY(hipEventElapsedTime(&ms, root.ev_k[0], root.ev_end));
        T.pipeline_us = ms * 1e3;
        HIP_TRY(hipEventElapsedTime(&ms, root.ev_k[n_brackets - 1], root.ev_end));
        T.tail_us = ms * 1e3 + host_tail_us;
    }
    T.n_gpus = P;
    T.q_batches = pl.nb;
    T.kv_splits = c.last_splits;
    const bool streamed0 = c.streamed && pl.r[0].stream.on;
    T.kv_chunks = streamed0 ? (int)pl.r[0].stream.end_tile.size() : (int)pl.r[0].chunks.size();
    T.streamed = streamed0 ? 1 : 0;
    T.host_convert_node = (c.hostcvt && E.hc) ? E.hc->placed_node() : -1;
    T.fused_launches = n_brackets / 2;
    T.plan = pl.qrows ? 1 : 0;
    T.merge = !pl.collectives ? 0 : (pl.merge_allreduce ? 2 : 1);
    T.virtual_ranks = E.virtual_ranks ? 1 : 0;
    T.enqueue_total_us = t_enqueued - t_enter;
    for (int g = 0; g < P && g < 16; ++g) T.enqueue_first_kernel_us[g] = c.first_kernel_us[g];
    T.egress = !pl.collectives ? 0 : (pl.egress_scatter ? 2 : 1);
    T.enqueue_threads = c.threaded ? P : 1;
    T.host_convert_threads = c.hostcvt ? E.hc->threads() : 0;
    T.compute_cus = pl.cus;
    T.stream_k = c.last_note.streamk;
    sdpa::format_launch_kernel(c.last_note, T.last_kernel, sizeof T.last_kernel);
    T.last_grid = c.last_note.grid;
    T.host_widen = c.widen ? 1 : 0;
    T.rccl_selftest = E.coll ? E.coll->selftested_ranks() : 0;
    if (c.tail_marked) {
        HIP_TRY(hipEventElapsedTime(&ms, root.ev_tail[0], root.ev_tail[1]));
        T.merge_us = ms * 1e3;
        HIP_TRY(hipEventElapsedTime(&ms, root.ev_tail[1], root.ev_tail[2]));
        T.reduce_us = ms * 1e3;
        HIP_TRY(hipEventElapsedTime(&ms, root.ev_tail[2], root.ev_end));
        T.egress_us = ms * 1e3;
    }
    return SDPA_OK;
}

int sdpa_kv_prefetch(const double *K, const double *V, int m, int n, int dk, int dv, int flags,
                     int k_rows_final, int v_rows_final) {
    SDPA_TRY(check_shape(K, K, V, V, m, n, dk, dv, flags, true));
    if (k_rows_final < 0 || v_rows_final < 0 || k_rows_final > n || v_rows_final > n) return SDPA_EINVAL;
    DeviceRestore restore;
    SDPA_TRY(lazy_init());
    Plan pl;
    make_plan(pl, m, n, dk, dv, flags);
    SDPA_TRY(check_plan(pl));
    if (host_cvt_mode() == 1) return SDPA_OK;   // host converts read the caller's arrays inside the compute call: nothing to stage ahead
    // (auto mode: a prefetched problem keeps its device converts -- the compute call finds PF.active)
    // the chunk layout is a function of environment knobs as well: a prefetch planned differently
    // from this call (other SDPA_KV_CHUNK_*, SDPA_QBATCH, rank count) starts over instead of
    // indexing its per-chunk state with the new plan's chunk numbers
    if (PF.active && PF.matches(K, V, m, n, dk, dv, flags)) {
        bool same = (int)PF.k_done.size() == pl.P && (int)PF.v_done.size() == pl.P;
        for (int g = 0; same && g < pl.P; ++g)
            same = PF.k_done[g].size() == pl.r[g].chunks.size() && PF.v_done[g].size() == pl.r[g].chunks.size();
        if (!same) PF.reset();
    }
    if (!PF.matches(K, V, m, n, dk, dv, flags)) {
        SDPA_TRY(ensure_buffers(pl));
        PF.reset();
        PF.active = true;
        PF.K = K; PF.V = V; PF.m = m; PF.n = n; PF.dk = dk; PF.dv = dv; PF.flags = flags;
        PF.k_done.resize(pl.P);
        PF.v_done.resize(pl.P);
        for (int g = 0; g < pl.P; ++g) {
            PF.k_done[g].assign(pl.r[g].chunks.size(), 0);
            PF.v_done[g].assign(pl.r[g].chunks.size(), 0);
        }
    }
    for (int g = 0; g < pl.P; ++g) {
        Rank &rk = E.r[g];
        const RankPlan &rp = pl.r[g];
        HIP_TRY(hipSetDevice(rk.dev));
        for (size_t c = 0; c < rp.chunks.size(); ++c) {
            const int end = rp.key_off + rp.chunks[c].k0 + rp.chunks[c].keys;     // global row past the chunk
            if (!PF.k_done[g][c] && end <= k_rows_final) {
                SDPA_TRY(stage_half(pl, rk, rp, K, (int)c, false));
                PF.k_done[g][c] = 1;
            }
            if (!PF.v_done[g][c] && end <= v_rows_final) {
                SDPA_TRY(stage_half(pl, rk, rp, V, (int)c, true));
                PF.v_done[g][c] = 1;
            }
        }
    }
    return SDPA_OK;
}

void *sdpa_host_alloc(size_t bytes) {
    if (bytes == 0 || sdpa::require_device() != SDPA_OK) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(HR.mu);
    HR.r.push_back({(const char *)p, bytes});
    return p;
}

void sdpa_host_free(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(HR.mu);
        for (size_t i = 0; i < HR.r.size(); ++i)
            if (HR.r[i].first == (const char *)p) { HR.r[i] = HR.r.back(); HR.r.pop_back(); break; }
        HR.probed.clear();
    }
    if (hipHostFree(p) != hipSuccess) (void)hipGetLastError();
}

// A caller that page-locks its arrays ITSELF (hipHostMalloc, a pinned torch tensor, hipHostRegister) says so: the range joins the ones
// sdpa_host_alloc() handed out and is used in place where that pays -- without the library asking the runtime about every pointer
// (ADVICE r5).  No device needed; the caller keeps the range page-locked until it forgets it.
int sdpa_host_declare_pinned(const void *p, size_t bytes) {
    if (!p || bytes == 0) return SDPA_EINVAL;
    std::lock_guard<std::mutex> lk(HR.mu);
    for (auto &e : HR.r)
        if (e.first == (const char *)p) { e.second = bytes; return SDPA_OK; }
    HR.r.push_back({(const char *)p, bytes});
    return SDPA_OK;
}

int sdpa_host_forget_pinned(const void *p) {
    if (!p) return SDPA_EINVAL;
    std::lock_guard<std::mutex> lk(HR.mu);
    for (size_t i = 0; i < HR.r.size(); ++i)
        if (HR.r[i].first == (const char *)p) { HR.r[i] = HR.r.back(); HR.r.pop_back(); return SDPA_OK; }
    return SDPA_EINVAL;
}

// The schedule sdpa_attention_f64 would run for this problem on `ranks` ranks, as one JSON object.
// Pure host arithmetic (no device, no engine): what the CPU test-suite checks the planner with.
int sdpa_plan_describe(int m, int n, int dk, int dv, int flags, int ranks, char *buf, size_t len) {
    if (!buf || len == 0 || ranks < 1 || ranks > sdpa::kMaxRanks) return SDPA_EINVAL;
    if (m < 0 || n < 0 || dk < 1 || dv < 1) return SDPA_EINVAL;
    sdpa::reload_launch_knobs();
    Plan pl;
    make_plan(pl, m, n, dk, dv, flags, ranks);
    std::string o;
    char t[320];
    snprintf(t, sizeof t, "{\"ranks\": %d, \"bf16\": %d, \"qrows\": %d, \"collectives\": %d, \"merge_allreduce\": %d, "
             "\"egress\": \"%s\", \"q_batch\": %d, \"q_batches\": %d, \"row_pieces\": %d, \"piece_min_rows\": %d, "
             "\"compute_cus\": %d, \"r\": [",
             pl.P, pl.bf16 ? 1 : 0, pl.qrows ? 1 : 0, pl.collectives ? 1 : 0, pl.merge_allreduce ? 1 : 0,
             !pl.collectives ? "own rows" : pl.egress_scatter ? "reduce-scatter" : "root", pl.B, pl.nb,
             pl.row_pieces, pl.piece_min_rows, pl.cus);
    o += t;
    for (int g = 0; g < pl.P; ++g) {
        const RankPlan &rp = pl.r[g];
        snprintf(t, sizeof t, "%s{\"key_off\": %d, \"key_cnt\": %d, \"row_off\": %d, \"row_cnt\": %d, \"n_slots\": %d, "
                 "\"ws_bytes\": %zu, \"piece_rows\": %d, \"chunks\": [", g ? ", " : "", rp.key_off, rp.key_cnt, rp.row_off,
                 rp.row_cnt, rp.n_slots, rp.ws_bytes, piece_rows_of(pl, std::min(pl.B, rp.row_cnt)));
        o += t;
        for (size_t c = 0; c < rp.chunks.size(); ++c) {
            snprintf(t, sizeof t, "%s[%d, %d, %d, %d]", c ? ", " : "", rp.chunks[c].k0, rp.chunks[c].keys,
                     rp.chunks[c].splits, rp.chunks[c].slot0);
            o += t;
        }
        // the streamed form of the first batch, where the shape allows it (taken when the call runs host converts):
        // splits of the ONE launch, tiles per split, the tile each group ends at, and the row ranges [first key, keys,
        // group] in the order they cross PCIe
        snprintf(t, sizeof t, "], \"stream\": {\"on\": %d, \"halves\": %d, \"rows_per_launch\": %d, \"splits\": %d, \"tiles_per_split\": %d, \"interleaved\": %d, \"q_with_group0\": %d, \"end_tile\": [",
                 rp.stream.on ? 1 : 0, rp.stream.halves, rp.stream.rows_per_launch, rp.stream.splits, rp.stream.tiles_per_split, rp.stream.interleaved ? 1 : 0,
                 rp.stream.q_with_group0 ? 1 : 0);
        o += t;
        for (size_t c = 0; c < rp.stream.end_tile.size(); ++c) {
            snprintf(t, sizeof t, "%s%d", c ? ", " : "", rp.stream.end_tile[c]);
            o += t;
        }
        o += "], \"entries\": [";
        for (size_t c = 0; c < rp.stream.entries.size(); ++c) {
            snprintf(t, sizeof t, "%s[%d, %d, %d]", c ? ", " : "", rp.stream.entries[c].k0, rp.stream.entries[c].keys,
                     rp.stream.entries[c].group);
            o += t;
        }
        o += "]}}";
    }
    {
        // the feed model of this plan: the three times (ms) and where the converts run for pageable / page-locked caller arrays
        const FeedModel f = feed_model(pl);
        snprintf(t, sizeof t, "], \"feed\": {\"cores\": %d, \"pool_threads\": %d, \"pool_GBps\": %.1f, \"t_host_ms\": %.3f, "
                 "\"t_link_ms\": %.3f, \"t_kernel_ms\": %.3f, \"pageable\": \"%s\", \"page_locked\": \"%s\"}}",
                 f.cores, f.threads, f.pool_Bps / 1e9, f.t_host * 1e3, f.t_link * 1e3, f.t_kernel * 1e3,
                 want_host_cvt(pl, true) ? "host" : "device", want_host_cvt(pl, false) ? "host" : "device");
        o += t;
    }
    if (o.size() + 1 > len) return SDPA_EINVAL;
    memcpy(buf, o.c_str(), o.size() + 1);
    return SDPA_OK;
}

int sdpa_prepare(int m, int n, int dk, int dv, int flags) {
    SDPA_TRY(check_shape(nullptr, nullptr, nullptr, nullptr, m, n, dk, dv, flags, false));
    DeviceRestore restore;
    SDPA_TRY(lazy_init());
    // 1. every device buffer the real call will use, at its real size
    Plan pl;
    make_plan(pl, m, n, dk, dv, flags);
    SDPA_TRY(check_plan(pl));
    SDPA_TRY(ensure_buffers(pl));
    // and the page-locked staging of the host converts (hundreds of MB: not inside a timed call) -- whenever the real
    // call COULD take them: it will if its arrays turn out to be pageable (the default no longer registers them)
    if (want_host_cvt(pl, !register_caller_arrays())) {
        SDPA_TRY(ensure_host_converter());
        const size_t vrow = pl.bf16 ? (size_t)dv * sizeof(unsigned short) : (size_t)pl.ldv * sizeof(float);
        size_t v_bytes = (size_t)n * vrow;
        if (pl.bf16)                      // (a streamed bf16 first batch stages the Vt images: padded dv rows x padded keys per rank)
            for (const RankPlan &rp : pl.r)
                if (rp.stream.on) v_bytes += (size_t)(sdpa::bf16_pad_dv(dv) - dv) * rp.key_cnt * 2 + (size_t)sdpa::bf16_pad_dv(dv) * 64;
        // (fp32 with equally wide images: K and V may share staging 0, group by group -- HostImages::pair)
        if (!E.hc->staging(0, (size_t)n * pl.ldk * pl.kv_elem * ((!pl.bf16 && pl.ldk == pl.ldv) ? 2 : 1)) || !E.hc->staging(1, v_bytes) ||
            !E.hc->staging(2, (size_t)m * pl.ldq * pl.q_elem))
            return SDPA_ENOMEM;
    }
    // ... and of the host-side widening: the warm-up call below is too small to take it, so without this the ONE timed
    // call of a CLI host paid a hipHostMalloc of m x dv floats and the creation of its "rows have landed" events
    // inside the timer
    if (want_host_widen(pl, !register_caller_arrays())) {
        SDPA_TRY(ensure_host_converter());
        if (!E.hc->staging(3, (size_t)m * dv * sizeof(float))) return SDPA_ENOMEM;
        const int need_ev = pl.nb * (kMaxSub + 1) + 2;
        for (int g = 0; g < pl.P; ++g) {
            Rank &rk = E.r[g];
            HIP_TRY(hipSetDevice(rk.dev));
            while ((int)rk.ev_w.size() < need_ev) {
                hipEvent_t e;
                HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                rk.ev_w.push_back(e);
            }
        }
    }
    // 2. one small call through the same code path: loads the code objects, sets the kernel
    //    attributes, creates the timing events (the kernel variants depend on dk, dv only)
    // (a problem whose first batch will run as the streamed launch warms up on the smallest one that does: 8192 rows --
    //  8 splits -- against 8192 keys, 0.3 ms of kernel; anything else on 256 rows)
    bool will_stream = false;
    for (const RankPlan &rp : pl.r) will_stream = will_stream || rp.stream.on;
    const int m_warm = will_stream ? 8192 : 256;
    const int m0 = m < m_warm ? m : m_warm, n0 = n < 8192 ? n : 8192;
    std::vector<double> q((size_t)m0 * dk, 0.25), k((size_t)n0 * dk, 0.5), v((size_t)n0 * dv, 1.0),
        r((size_t)m0 * dv);
    const sdpa_timing keep = E.last;
    // (the start-up PROBE of the streamed launch: with a short wait bound -- a runtime on which the copies do not land beside the
    //  launch is found out here, in ~0.2 s and outside any timer, and the process keeps the launch-per-chunk schedule)
    if (will_stream) E.stream_timeout_ms_once = sdpa_debug_pos("stream_probe_ms", 200);
    const bool was_off = E.stream_off;
    const int rc = sdpa_attention_f64(q.data(), k.data(), v.data(), r.data(), m0, n0, dk, dv, flags);
    E.last = keep;
    if (rc == SDPA_OK && E.stream_off && !was_off) {        // the probe failed: size the buffers of the schedule the real call will take
        make_plan(pl, m, n, dk, dv, flags);
        SDPA_TRY(ensure_buffers(pl));
    }
    // 3. the core clock.  From idle this part needs ~20 ms of continuous matrix work to reach its plateau
    //    (profiles/r02/short_step_clock_ramp.log: the same launch 290 us at the start, 253 us from then on): a host that
    //    makes ONE timed call (both CLIs) would time it on the ramp -- 10-20 ms cold against 9 ms warm at the metric
    //    shape (profiles/r03/cli_one_shot_timing.log).  So the real launch shape runs on operand-like images for about
    //    $SDPA_PREPARE_WARM_MS (default 60) here, outside any timer, the way the reference does MPI_Init and its
    //    transport set-up before it starts the clock (attention-mpi.c:504, :519).  0 = off.
    const int warm_ms = rc != SDPA_OK ? 0 : getenv("SDPA_PREPARE_WARM_MS") ? atoi(getenv("SDPA_PREPARE_WARM_MS")) : 60;
    if (warm_ms > 0) {
        make_plan(pl, m, n, dk, dv, flags);
        SDPA_TRY(ensure_buffers(pl));          // (the small call re-pointed the V image behind ITS K image)
        for (int g = 0; g < pl.P; ++g) {
            Rank &rk = E.r[g];
            const RankPlan &rp = pl.r[g];
            const int rows = std::min(pl.B, rp.row_cnt);
            if (rows <= 0 || rp.key_cnt <= 0) continue;
            HIP_TRY(hipSetDevice(rk.dev));
            // (operand-LIKE data, not zeros: MFMAs on zeros do not bring the part out of its light-load power state, and the one timed
            //  call of a CLI host then ran its kernel 1.4 ms slower however long this loop was -- launch_fill_pattern says what was measured)
            if (sdpa_debug_int("prepare_zero", 0) != 0) {
                HIP_TRY(hipMemsetAsync(rk.qf[0].p, 0, (size_t)rows * pl.ldq * pl.q_elem, rk.s_run));
                HIP_TRY(hipMemsetAsync(rk.kf.p, 0, (size_t)rp.key_cnt * pl.ldk * pl.kv_elem, rk.s_run));
                HIP_TRY(hipMemsetAsync(rk.vf.p, 0, pl.bf16 ? (size_t)sdpa::bf16_pad_dv(dv) * sdpa::bf16_pad_n(rp.key_cnt) * sizeof(unsigned short)
                                                           : (size_t)rp.key_cnt * pl.ldv * sizeof(float), rk.s_run));
            } else {
            HIP_TRY(sdpa::launch_fill_pattern(rk.qf[0].p, (size_t)rows * pl.ldq * pl.q_elem, pl.bf16, pl.bf16 ? 1.44269504f / sqrtf((float)dk) : 1.0f, rk.s_run));
            HIP_TRY(sdpa::launch_fill_pattern(rk.kf.p, (size_t)rp.key_cnt * pl.ldk * pl.kv_elem, pl.bf16, 1.0f, rk.s_run));
            HIP_TRY(sdpa::launch_fill_pattern(rk.vf.p, pl.bf16 ? (size_t)sdpa::bf16_pad_dv(dv) * sdpa::bf16_pad_n(rp.key_cnt) * sizeof(unsigned short)
                                                                : (size_t)rp.key_cnt * pl.ldv * sizeof(float), pl.bf16, 1.0f, rk.s_run));
            }
            // (the launch count comes from the MEASURED duration of the first launch, not from an MFMA-rate estimate --
            //  the VALU-only kernels of dk > 1024 are 10-100x slower than any such estimate and would warm for seconds)
            const int sp = pick_splits(pl, rows, rp.key_cnt);
            SDPA_TRY(launch_fused(pl, rk, rp, 0, rows, 0, rows, 0, rp.key_cnt, sp, -1));
            HIP_TRY(hipStreamSynchronize(rk.s_run));
            const double t0 = now_us();
            SDPA_TRY(launch_fused(pl, rk, rp, 0, rows, 0, rows, 0, rp.key_cnt, sp, -1));
            HIP_TRY(hipStreamSynchronize(rk.s_run));
            const double t_launch = std::max(1e-6, (now_us() - t0) * 1e-6);
            const int reps = (int)std::min(2000.0, std::max(0.0, warm_ms * 1e-3 / t_launch - 2.0));
            for (int i = 0; i < reps; ++i) SDPA_TRY(launch_fused(pl, rk, rp, 0, rows, 0, rows, 0, rp.key_cnt, sp, -1));
        }
        for (int g = 0; g < pl.P; ++g) {
            HIP_TRY(hipSetDevice(E.r[g].dev));
            HIP_TRY(hipStreamSynchronize(E.r[g].s_run));
        }
    }
    //    LAST (round 6): the state this buys does not survive the small call above -- a streamed launch whose workgroups mostly wait --
    //    when it runs in between: the first timed call's kernel 9.2-9.7 ms with the warm-up in front of the small call, 8.1-8.2 with four or
    //    more dense launches right in front of the timed call (profiles/r06/first_call_warmup.log).  The converter pool's threads, asleep
    //    for those 60 ms, are woken by a round of small conversions afterwards (round 5 measured a timed call that has to wake all of
    //    them from deep sleep at +1 ms of head).
    if (warm_ms > 0 && E.hc) {          // wake the converter pool: every thread gets a few rows to convert
        const int th = std::max(1, E.hc->threads());
        std::vector<double> src((size_t)th * 2 * 1024, 0.5);
        std::vector<float> dst(src.size());
        E.hc->begin();
        for (int t = 0; t < 2 * th; ++t) (void)E.hc->submit(src.data() + (size_t)t * 1024, dst.data() + (size_t)t * 1024, 8, 128, 128, sdpa::kCvtF32, 1.0);
        E.hc->kick();
        E.hc->finish();
    }
    return rc;
}

}  // extern "C"
