/*
 * sdpa_hip.h -- C ABI of the MI355X (gfx950) scaled-dot-product-attention engine.
 *
 * This is the drop-in boundary for the one hot path of the reference
 * (paths relative to the reference tree):
 *
 *     attention()                  attention.c:20-21      (serial, fp64)
 *     attention()                  attention-mpi.c:191-192 (K/V-sharded, fp32 compute)
 *       online_softmax_attention   attention-mpi.c:168-189
 *       dot_avx512 / axpy_avx512 / memset_zero_scale   :103-166
 *       cvt_d2f_avx512 / cvt_f2d_avx512                :31-101
 *       owner_count / owner_disp                       :19-27
 *       two-phase merge + reduce                       :340-399
 *
 * Everything here is plain C: pointers, ints, sizes.  No C++ or torch types
 * cross the boundary and no C++ exception escapes it.  Every entry point
 * returns 0 on success or a negative SDPA_E* code (sdpa_strerror() names it).
 * There is NO CPU fallback: without a usable gfx950 device every compute
 * entry point fails with SDPA_ENODEV.
 *
 * Two layers:
 *   1. host level  -- sdpa_attention_f64(): host fp64 in / host fp64 out, the
 *      body of the reference's attention(); drives 1..8 GPUs from one process.
 *   2. device level -- sdpa_dev_*(): the individual stages on DEVICE pointers
 *      and a caller-supplied hipStream_t (passed as void*), for hosts that own
 *      device memory and collectives themselves (one process per GPU with RCCL
 *      through torch.distributed: bench.py, the package's Python host).
 *
 * Environment (the WHOLE list; read once per call on the calling thread; anything else lives in $SDPA_DEBUG):
 *   SDPA_GPUS=N              use N devices (default: every visible one that passes the RCCL self-test)
 *   SDPA_VIRTUAL_GPUS=P      P loopback ranks on ONE device (the P > 1 pipeline on a one-GPU machine; tests)
 *   SDPA_PRECISION=bf16      as SDPA_F_BF16            SDPA_PLAN=qrows    as SDPA_F_PLAN_QROWS
 *   SDPA_MERGE=allreduce     as SDPA_F_MERGE_ALLREDUCE  SDPA_EGRESS=root|scatter   who copies result rows home (P > 1)
 *   SDPA_COMM_CUS=k          compute units left to RCCL's kernels when several ranks are driven (default 16)
 *   SDPA_RCCL_SELFTEST_TIMEOUT_S   deadline of sdpa_init()'s self-test (default 60)
 *   SDPA_QBATCH=rows         query rows per batch (default: the whole array up to 32768)
 *   SDPA_STREAMED=0|1        the first batch as ONE persistent launch fed by the copy engine (default: when the start-up probe passed)
 *   SDPA_STREAM_TIMEOUT_MS   how long such a launch waits for a chunk before it gives up and the call re-runs chunked (default 500)
 *   SDPA_HOST_CVT=0|1|auto   fp64 -> operand converts on the device / on host threads / by the feed model (default auto)
 *   SDPA_HOST_CVT_THREADS    size of the converter pool (default 2 x the cores the process may use, <= 128)
 *   SDPA_HOST_WIDEN=0|1|auto result rows widened to fp64 on the device / on host threads (default auto)
 *   SDPA_HOST_REGISTER=1     page-lock the caller's arrays for the call (off: INTEGRATION.md says why)
 *   SDPA_PREPARE_WARM_MS     sdpa_prepare()'s clock warm-up (default 60, 0 = off)
 *   SDPA_CLI_PREFETCH=1      the CLI hosts hand K/V rows over while they are still reading the file
 *   SDPA_VERBOSE=1           one timing line per call on stderr
 *   SDPA_DEBUG="name=value,..."    every test / tuning / experiment knob (csrc/sdpa_debug.h lists them; not an interface)
 *   (the package's ctypes loader, bench.py and tools/ also read SDPA_HIP_LIB: the path of a differently built copy of this library)
 */
#ifndef SDPA_HIP_H
#define SDPA_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDPA_API __attribute__((visibility("default")))

/* error codes */
#define SDPA_OK        0
#define SDPA_EINVAL   -1   /* bad argument (null pointer, non-positive dim, bad ld) */
#define SDPA_ENODEV   -2   /* no usable HIP device / engine not initialised        */
#define SDPA_EHIP     -3   /* a HIP runtime call failed (message on stderr)         */
#define SDPA_ERCCL    -4   /* an RCCL call failed                                    */
#define SDPA_ENOMEM   -5   /* device or pinned-host allocation failed                */
#define SDPA_EUNSUP   -6   /* shape not supported by this build                      */

/* flags for sdpa_attention_f64 / sdpa_prepare */
#define SDPA_F_DEFAULT         0
#define SDPA_F_NO_PIPELINE     1   /* one Q batch, one K/V chunk, no overlap (debug)            */
#define SDPA_F_BF16            2   /* bf16-input MFMA path (fp32 accumulate/softmax);           */
                                   /* also selected by $SDPA_PRECISION=bf16                     */
#define SDPA_F_PLAN_QROWS      4   /* multi-GPU plan: shard the QUERY rows, replicate K/V, no   */
                                   /* merge collective (rows are independent, attention.c:28);  */
                                   /* also $SDPA_PLAN=qrows.  Default: K/V rows sharded         */
                                   /* (attention-mpi.c:19-27) + the merge of :340-380           */
#define SDPA_F_MERGE_ALLREDUCE 8   /* K/V plan: the reference's literal two-phase merge,        */
                                   /* all-reduce(MAX) + all-reduce(SUM) (attention-mpi.c:342,   */
                                   /* :354); also $SDPA_MERGE=allreduce.  Default: ONE          */
                                   /* all-gather of the (lmax,lsum) pairs -- same algebra.      */
                                   /* Either way the merged rows are reduce-SCATTERED, every    */
                                   /* rank copies its share home ($SDPA_EGRESS=root: the        */
                                   /* reference's reduce to rank 0, attention-mpi.c:380)        */

/* Breakdown of the last sdpa_attention_f64(), microseconds.  The call only ENQUEUES work and
 * waits once at the end, so the stage figures are device-side intervals (HIP events on GPU 0)
 * and they overlap each other; total_us is the host wall clock of the call.                    */
struct sdpa_timing {
    double total_us;      /* entry -> exit of sdpa_attention_f64 (host clock)                   */
    double kv_stage_us;   /* first H2D -> last K/V chunk converted (runs UNDER the kernels)     */
    double pipeline_us;   /* first fused kernel start -> last result byte on the host           */
    double kernel_us;     /* sum of fused-kernel launch durations on GPU 0                      */
    int    n_gpus;        /* ranks the call used (GPUs, or virtual ranks)                       */
    int    q_batches;     /* Q batches the pipeline ran                                         */
    int    kv_splits;     /* in-GPU K/V splits of the last fused launch                         */
    /* -- fields added in 0.2 (appended: older readers of the struct stay valid) -------------- */
    double register_us;   /* page-locking the caller's arrays (host clock, inside total_us);     */
                          /* 0 unless $SDPA_HOST_REGISTER=1 (round 4: off by default)            */
    double head_us;       /* entry -> first fused kernel starts (what is NOT overlapped at the  */
                          /* front: registration, Q batch 0 and K/V chunk 0 over PCIe)          */
    double tail_us;       /* last fused kernel ends -> exit (merge, collectives, last D2H)      */
    int    kv_chunks;     /* K/V chunks the first Q batch streamed through (1 = not streamed)   */
    int    fused_launches;/* fused-kernel launches on GPU 0                                     */
    int    plan;          /* 0 = K/V rows sharded, 1 = query rows sharded                       */
    int    merge;         /* 0 = none (one rank), 1 = all-gather, 2 = two all-reduces           */
    int    virtual_ranks; /* 1 = the ranks are loopback ranks on one device                     */
    /* -- fields added in 0.3 (appended) ------------------------------------------------------ */
    double enqueue_total_us;            /* entry -> every rank's work and every collective tail   */
                                        /* is enqueued (host clock; the GPUs are long at work)    */
    double enqueue_first_kernel_us[16]; /* entry -> rank r's FIRST fused launch is enqueued (host */
                                        /* clock).  With P > 1 every rank has its own enqueue     */
                                        /* thread: the spread over ranks is thread wake-up skew,  */
                                        /* not one rank's whole batch of API calls per rank       */
    int    egress;        /* how finished rows go home: 0 = from the rank that computed them,    */
                          /* 1 = reduce to the root (attention-mpi.c:380), 2 = reduce-scatter,   */
                          /* every rank widens and copies its rows over its own PCIe link        */
    int    enqueue_threads; /* host threads that enqueued (1 = the calling thread only)          */
    int    host_convert_threads; /* $SDPA_HOST_CVT=1: host threads that converted fp64 -> operand  */
                          /* images (attention-mpi.c:224-225's placement); 0 = the device converts */
    /* -- fields added in ABI 4 (appended) ---------------------------------------------------- */
    int    compute_cus;   /* compute units the fused kernels' stream may use (256 - $SDPA_COMM_CUS)  */
    int    stream_k;      /* 1 = the last fused launch used the stream-K work distribution           */
    int    host_widen;    /* 1 = fp32 rows came home and host threads widened them to fp64           */
                          /* (cvt_f2d_avx512's placement, attention-mpi.c:373,:396)                  */
    int    rccl_selftest; /* 1 = the engine's RCCL collectives passed their known-answer self-test   */
                          /* on this many ranks at creation (0 = loopback ranks / one rank)          */
    double merge_us;      /* rank 0, last batch: all-gather/all-reduce + merge kernel (comm stream)  */
    double reduce_us;     /* rank 0, last batch: reduce / reduce-scatter of the contributions        */
    double egress_us;     /* rank 0, last batch: widen + D2H of its rows                             */
    /* -- fields added in ABI 5 (appended) ---------------------------------------------------- */
    char   last_kernel[96]; /* rank 0's last fused launch as rocprofv3 names it, e.g.                 */
                          /* "sdpa::fused_pipelined_kernel<128,128,0,0>" (what LAUNCHED, recorded by  */
                          /* the launcher -- not a prediction)                                        */
    int    last_grid;     /* its workgroups                                                          */
    int    streamed;      /* 1 = the first Q batch ran as ONE persistent launch that followed the K/V */
                          /* chunks as they arrived (round 5), 0 = one launch per chunk               */
    int    host_convert_node; /* $SDPA_DEBUG host_cvt_pin=1 (opt-in): the NUMA node the converter pool's threads */
                          /* were confined to for this call (where the source arrays' pages live);     */
                          /* -1 = every CPU the process may use (the default)                          */
};

/* ---- lifecycle ---------------------------------------------------------- */

/* Create the engine on `n_gpus` devices (0 = every visible device).  Separate
 * from the compute call so a bench can keep one-time HIP/RCCL start-up out of
 * the timed region (the reference's timer brackets attention() itself,
 * attention.c:179-182).  sdpa_attention_f64() creates the engine lazily when
 * none exists: see sdpa_init_default() below (every visible device whose RCCL
 * transport passes the self-test; $SDPA_GPUS=N forces a count).
 * $SDPA_VIRTUAL_GPUS=P makes the engine P logical ranks that all live on
 * device 0 (own streams and buffers each, loopback collectives): the P > 1
 * pipeline on a one-GPU machine.  $SDPA_DEBUG force_collectives=1 runs the merge
 * collectives even with one rank (a one-rank RCCL communicator).
 *
 * Threading: the engine is one process-wide object; the host-level entry
 * points are NOT thread-safe and not re-entrant (the reference's caller is
 * single-threaded, attention.c:179-182).  They restore the calling thread's
 * current HIP device before returning.                                        */
SDPA_API int sdpa_init(int n_gpus);
/* The engine sdpa_attention_f64() creates lazily, created now: $SDPA_GPUS devices when set; otherwise EVERY visible
 * device, on the evidence of this node -- creating an engine on P > 1 devices runs each collective of the pipeline
 * once over RCCL on known data (a deadline of $SDPA_RCCL_SELFTEST_TIMEOUT_S, default 60 s): passed = P GPUs; failed
 * with an error = one line on stderr and ONE GPU; did not finish = SDPA_ERCCL (set SDPA_GPUS=1).  A no-op when an
 * engine exists.  sdpa_engine_ranks(): ranks of the current engine (0 = none).                                    */
SDPA_API int sdpa_init_default(void);
SDPA_API int sdpa_engine_ranks(void);
SDPA_API void sdpa_shutdown(void);
SDPA_API int sdpa_device_count(void);          /* visible HIP devices, <0 on error */
SDPA_API const char *sdpa_strerror(int code);
/* "sdpa-hip <abi> (gfx950, ...; hipcc <version>; src <sha256/16 of the kernel sources>)": the compiler the
 * hand-scheduled kernels were built with and the sources they were built from travel with the library (the
 * GPU boxes run a prebuilt .so; tests/test_kernel_isa.py's guarantees hold for that compiler only).        */
SDPA_API const char *sdpa_version(void);
/* Bumped whenever a struct of this header changes size or a documented behaviour changes; hosts compare it
 * with the SDPA_ABI_VERSION they were compiled against (the package's ctypes loader and both C hosts do). */
#define SDPA_ABI_VERSION 6
SDPA_API int sdpa_abi_version(void);

/* The launch paths read their knobs ($SDPA_DEBUG: streamk, split_merge)
 * from ONE snapshot, taken at first use and by every host-level entry point on the
 * calling thread: the launchers run on the engine's enqueue threads, and the C environment must not be
 * read there while the application may setenv().  A device-level host that changes one of these knobs
 * between launches calls this (from the thread that changed it, with no launch in flight elsewhere).      */
SDPA_API void sdpa_reload_env(void);

/* ---- host level: replaces the body of attention() ------------------------ */

/* result[m x dv] = softmax(Q K^T / sqrt(dk)) V.
 * Q[m x dk], K[n x dk], V[n x dv], result[m x dv]: dense row-major fp64 host
 * arrays owned by the caller (attention.c:14-21); inputs are not written,
 * result is fully overwritten, no pointer is retained.
 * Compute is fp32 (fp64 inputs rounded to nearest-even as cvt_d2f_avx512,
 * attention-mpi.c:31-64; scale = 1/sqrtf((float)dk), :208), output widened to
 * fp64 (:373,:396).  K/V rows are sharded over the engine's GPUs with
 * owner_count/owner_disp (:19-27) and merged with the algebra of :340-380
 * (flags choose the collective schedule).  The first Q batch streams the K/V
 * shard host->device in groups of rows and computes as they land: for the
 * fp32 shapes with dk, dv in (32, 128] and the bf16 shapes with dv > 256 ONE
 * persistent launch whose workgroups wait, in the kernel, for the ready word
 * of the rows they are about to read (raised by the copy engine behind the
 * rows; bf16: V travels as column ranges of the transposed image the host
 * writes, sdpa_host_cvt_vt; sdpa_timing.streamed = 1; the same triples, bit
 * for bit, as sdpa_dev_shard_partial_f32 / _bf16 on the resident shard;
 * $SDPA_STREAMED=0: one launch per K/V chunk, as for every other shape).
 * Every shape has a kernel: any dk, dv in fp32 (dk > 1024 on a VALU-only
 * kernel); SDPA_F_BF16 beyond the bf16 kernels' dims (dk <= 512, dv <= 1024,
 * 4 GiB of Vt per rank) runs the fp32 path and says so on stderr.
 * Numerical range: the fp32 kernels rescale their accumulators lazily (only
 * when a row maximum rises by more than 2^24), which spends that much of
 * fp32's exponent headroom: the un-normalised contrib of a row overflows for
 * |V|*n above ~2^104, where the reference's eager rescale would not.  The
 * bf16 duo and tandem kernels keep NO reference exponent (P = 2^score) while
 * the row's sum of 2^score stays inside [2^-80, 2^80] -- roughly every score
 * within |q.k/sqrt(dk)| <= 55; rows outside are recomputed by the rescaling
 * kernel, no loss: there contrib overflows for |V| above ~2^47.            */
SDPA_API int sdpa_attention_f64(const double *Q, const double *K, const double *V,
                                double *result, int m, int n, int dk, int dv,
                                int flags);
/* Copies min(size, sizeof(struct sdpa_timing)) bytes: a caller compiled against an older, shorter struct
 * passes ITS sizeof and is never written past it (fields are only ever appended).  sdpa_last_timing(out)
 * is sdpa_last_timing_sized(out, sizeof of THIS header's struct) and is kept for round-2/3 binaries only. */
SDPA_API int sdpa_last_timing_sized(struct sdpa_timing *out, size_t size);
SDPA_API int sdpa_last_timing(struct sdpa_timing *out);

/* Optional: size the engine for one problem before the timed call -- allocates every device
 * buffer sdpa_attention_f64(m,n,dk,dv,flags) will use and runs a small problem of the same
 * dk, dv through the same code path so that code objects are loaded, then keeps the matrix
 * cores busy for ~60 ms on zeroed operands ($SDPA_PREPARE_WARM_MS, 0 = off): from idle the core
 * clock needs about that long to reach its plateau, and a host that makes ONE timed call would
 * time it on the ramp.  The analogue of the reference doing MPI_Init and its transport set-up
 * outside the timer (attention-mpi.c:10-17, :504); sdpa_attention_f64 works without it, the first
 * call is just slower.                                                                          */
SDPA_API int sdpa_prepare(int m, int n, int dk, int dv, int flags);

/* The schedule sdpa_attention_f64 would run for this problem on `ranks` ranks (1..16), as one JSON
 * object in buf: Q batch size and count, row pieces, the compute units the fused launches may use
 * ("compute_cus": the chip, or 16 fewer when the call leaves workgroup slots to the comm streams),
 * and per rank its K/V rows (owner_count /
 * owner_disp, attention-mpi.c:19-27) or query rows (SDPA_F_PLAN_QROWS), the streamed K/V chunks
 * [first key, keys, in-launch splits, first slot] and the scratch it needs.  Reads the same
 * environment knobs as the call itself.  Pure host arithmetic: needs no GPU and no engine.
 * SDPA_EINVAL when buf is too small.                                                          */
SDPA_API int sdpa_plan_describe(int m, int n, int dk, int dv, int flags, int ranks, char *buf,
                                size_t len);

/* Optional: for hosts that produce or READ K and V incrementally (the CLI's file reader with
 * SDPA_CLI_PREFETCH=1, attention.c:100-121).  Says that rows [0, k_rows_final) of K and
 * [0, v_rows_final) of V are final in host memory; the engine starts moving every K/V chunk that
 * is complete to the device(s) and converting it, while the host carries on reading.  The NEXT
 * sdpa_attention_f64() with the same K, V, m, n, dk, dv, flags (and the same environment knobs)
 * skips what is already staged; any other call voids the prefetch.  Call it again as more rows
 * become final (counts only grow).  The rows announced must not change, and K / V must stay valid,
 * until that compute call returns.  Note for benchmarks: a compute call that follows a prefetch
 * does not contain the K/V transfer -- the reference's timed region does (attention-mpi.c:210-266).
 * Call sdpa_prepare() BEFORE the first prefetch, not after (its warm-up call voids it).        */
SDPA_API int sdpa_kv_prefetch(const double *K, const double *V, int m, int n, int dk, int dv,
                              int flags, int k_rows_final, int v_rows_final);

/* Optional: page-locked host memory for the caller's Q/K/V/result arrays -- what the reference's
 * read_matrix() mallocs (attention.c:84-90) and main() frees (:191-194).  Arrays allocated here
 * are used in place by sdpa_attention_f64 (fp64 straight over PCIe at the full rate, device
 * converts).  Any other host pointer is accepted as well: since round 4 a pageable array is NEVER
 * registered behind the caller's back ($SDPA_HOST_REGISTER=1 opts in; INTEGRATION.md says why
 * not to) -- its rows go through the library's page-locked staging on host threads instead.
 * Returns NULL when there is no usable device or the allocation fails.  SURVEY.md 8(f)-2.       */
SDPA_API void *sdpa_host_alloc(size_t bytes);
SDPA_API void  sdpa_host_free(void *p);
/* A caller whose arrays are page-locked by OTHER means (hipHostMalloc, a pinned torch tensor, hipHostRegister of its own) declares the
 * range [p, p + bytes): it is then treated like a sdpa_host_alloc range.  Why declare: since round 5 the library takes every pointer it
 * does not know for pageable WITHOUT asking the runtime (hipPointerGetAttributes on a plain malloc makes ROCm 7 log an error-level line
 * per pointer); results are the same either way, an undeclared pinned array merely travels through the staging like a pageable one.
 * The range must stay page-locked until sdpa_host_forget_pinned(p).  No device needed.  SDPA_EINVAL: null pointer / zero size / a
 * pointer that was never declared.                                                                                                  */
SDPA_API int   sdpa_host_declare_pinned(const void *p, size_t bytes);
SDPA_API int   sdpa_host_forget_pinned(const void *p);

/* The host-side converter: `rows` rows of fp64 -> rows of an operand image, on the calling thread, with
 * the device converters' roundings bit for bit.  kind 0: float rows of `ld` floats, pad columns zero
 * (cvt_d2f_avx512, attention-mpi.c:31-64: vcvtpd2ps, 8 doubles at a time, where the CPU has AVX-512);
 * kind 1: bf16 rows, bf16((float)(x * mult)), both roundings to nearest even; kind 2 (ABI 6): rows of the TILED K image of a
 * dv > 256 shape -- kind 1's values with 16-byte chunk c of row r (counted from dst, which must be an image row that is a multiple
 * of 16) stored at chunk position c ^ (r & min(15, ld/8 - 1)), ld = the padded dk.  flags bit 0: the plain C
 * rows instead of the AVX-512 ones; bit 1 / bit 2: streaming (non-temporal) stores for line-aligned
 * destination rows on / off (neither: the pool's default, streaming) -- the same bytes either way.  This is what $SDPA_HOST_CVT=1 runs on a pool of host threads inside
 * sdpa_attention_f64 (the reference's own placement of the converts, :224-225, :303); it needs no GPU.  */
SDPA_API int sdpa_host_cvt_rows(const double *src, void *dst, long rows, int cols, int ld, int kind,
                                double mult, int flags);

/* The host-side converter of the bf16 kernels' TRANSPOSED V image (round 5: what the persistent bf16 launch of the first Q batch is
 * fed with): `keys` rows of V (fp64, `cols` columns) become the key positions [0, keys_pad) of an image whose rows are `ldt`
 * elements apart -- dst[c * ldt + p] = bf16((float)V[j][c]) with p = j with bits 2 and 3 swapped (each 16-key group is stored
 * 0-3, 8-11, 4-7, 12-15: one MFMA lane's eight keys are 16 contiguous bytes), zero for positions behind the last key (keys_pad: a
 * multiple of 32 >= keys), zero rows for c in [cols, cols_pad).  cols > 256 (ABI 6): the TILED image instead -- dst = the block of
 * the first key's tile, [keys_pad/32 tiles][cols_pad/512][512][32], ldt unused (see the device-level section).  The device converter's image bit for bit; flags as
 * sdpa_host_cvt_rows; threads <= 1: on the calling thread, threads > 1: on a pool of that many threads, in work items of whole
 * 32-key tiles -- the way sdpa_attention_f64 runs it.  The reference converts V on the host as well (cvt_d2f_avx512 at attention-mpi.c:225); needs no GPU.   */
SDPA_API int sdpa_host_cvt_vt(const double *src, unsigned short *dst, long keys, long keys_pad, int cols, int cols_pad,
                              long ldt, int threads, int flags);

/* The host-side widening of result rows: dst[i] = (double)src[i] for i < n (cvt_f2d_avx512, attention-mpi.c:68-101,
 * called on the root at :373 / :396), exact.  threads <= 1: on the calling thread; threads > 1: on a pool of that many
 * threads plus the calling one, the way $SDPA_HOST_WIDEN runs it inside sdpa_attention_f64 (fp32 rows cross PCIe, the
 * host widens them into `result`).  flags bit 0: the plain C loop instead of the AVX-512 one.  Needs no GPU.     */
SDPA_API int sdpa_host_widen(const float *src, double *dst, size_t n, int threads, int flags);

/* K/V row partition, attention-mpi.c:19-27. */
SDPA_API int sdpa_owner_count(int n, int size, int rank);
SDPA_API int sdpa_owner_disp(int n, int size, int rank);

/* ---- device level -------------------------------------------------------- */
/* All pointers below are DEVICE pointers on the current HIP device; `stream`
 * is a hipStream_t (NULL = the default stream).  Calls only enqueue work.
 * Leading dimensions are in elements, must be multiples of 4 and >= the column
 * count; columns [cols, ld) of every fp32 matrix handed to
 * sdpa_dev_shard_partial_f32 must be zero (sdpa_dev_cvt_d2f writes them so).
 * Operand base pointers must be 16-byte aligned (SDPA_EINVAL otherwise).        */

/* A stream for the fused kernels that leaves `reserve_cus` compute units' worth of the chip (rounded up to a
 * multiple of the 8 XCDs) to other streams' kernels; 0 = an ordinary non-blocking stream.  Why: a fused launch
 * otherwise holds every workgroup slot of every CU until its last workgroup ends, so a collective (RCCL) or merge
 * kernel that becomes ready while it runs cannot start -- whatever its stream or priority.  A host that wants
 * batch b's reduce to run UNDER batch b+1's fused kernel (attention-mpi.c:364-380) launches the fused kernels on
 * such a stream (the C host: $SDPA_COMM_CUS, default 16 when it drives several ranks; bench.py: --reserve-cus).
 * The reservation is made by grid size: the stream is an ordinary non-blocking stream that this library remembers
 * as having (256 - reserve) CUs, and sdpa_dev_shard_partial_f32 sizes its stream-K grid by that -- 2 x reserve of
 * the 512 workgroup slots stay free on CUs that hold one fused workgroup.  It costs the fused kernel reserve/256
 * of its rate and no more.  (Head dims whose kernels have no stream-K form -- dk > 128, bf16 -- ignore it.)      */
SDPA_API int sdpa_dev_stream_create(int reserve_cus, void **stream);
/* The calling thread's LAST fused launch through sdpa_dev_shard_partial_f32 / _bf16, as one JSON object:
 * {"kernel": "sdpa::fused_pipelined_kernel<128,128,0,0>", "grid": 512, "splits": 2, "stream_k": 0, "rows": m,
 *  "keys": n_local} -- the kernel name is the one rocprofv3 prints, recorded by the launcher itself, so a bench
 * line or a profile reader names what ran (a stream with a reservation launches the stream-K form
 * fused_pipelined_sk_kernel<DK,DV>).  SDPA_EINVAL when buf is too small.                                       */
SDPA_API int sdpa_dev_last_launch(char *buf, size_t len);
SDPA_API int sdpa_dev_stream_destroy(void *stream);

/* The leading dimension to give the fp32 images of a matrix with d columns (and the contrib rows
 * of a dv-column result): d in (32, 256] padded to 64 / 128 / 256, d <= 32 rounded up to 4, beyond
 * 256 to 12 / 4 / 12 / 8 for d <= 384 / 512 / 768 / more (whole lane runs of the dk-split kernels'
 * matched dv slices; 384, 512, 768, 1024 stay as they are).  Images of those widths run the fastest
 * kernel for the head dims; any other ld >= d (multiple of 4, pad columns zero) is accepted and
 * takes the any-shape kernels / the 128-wide dv slices.                                          */
SDPA_API int sdpa_dev_dense_ld(int d);

/* fp64 -> fp32, round-to-nearest-even; dst[r*ld + c], pad columns zeroed.
 * Replaces cvt_d2f_avx512 (attention-mpi.c:31-64).                           */
SDPA_API int sdpa_dev_cvt_d2f(const double *src, float *dst, long rows, int cols,
                              int ld, void *stream);
/* The same for up to three matrices in ONE launch (ABI 6: a short call's K, V and Q -- three launches' gaps are 3 % of config 2's
 * step): arrays of `count` sources / destinations / rows / cols / lds; every image is sdpa_dev_cvt_d2f's bit for bit.           */
SDPA_API int sdpa_dev_cvt_d2f_batch(int count, const double *const *src, float *const *dst, const long *rows, const int *cols,
                                    const int *ld, void *stream);
/* fp32 -> fp64 (exact).  Replaces cvt_f2d_avx512 (:68-101).                   */
SDPA_API int sdpa_dev_cvt_f2d(const float *src, int ld, double *dst, long rows,
                              int cols, void *stream);

/* Number of in-GPU K/V splits (slabs of partial triples) the fused kernel uses for this shape on a
 * stream that owns the whole chip, and the scratch a launch needs on ANY stream -- one created with
 * sdpa_dev_stream_create(reserve > 0) cuts the work differently (stream-K over fewer workgroup slots)
 * and may use a slab more; the byte count covers both (0 bytes when the answer is 1 split everywhere). */
SDPA_API int    sdpa_dev_kv_splits(int m, int n_local, int dk, int dv);
SDPA_API size_t sdpa_dev_workspace_bytes(int m, int n_local, int dk, int dv);

/* The fused kernel: online_softmax_attention (attention-mpi.c:168-189) for ALL
 * m query rows against one K/V shard of n_local rows:
 *   contrib[m x dv] (ld ldo) = sum_j exp(s_ij - lmax_i) V_j     (UN-normalised)
 *   lmax[m] = max_j s_ij,  lsum[m] = sum_j exp(s_ij - lmax_i),
 *   s_ij = dot(Q_i, K_j) / sqrtf(dk).
 * n_local == 0 gives contrib = 0, lmax = -inf, lsum = 0 (:172-173).
 * `workspace` must hold sdpa_dev_workspace_bytes() bytes (may be NULL if 0). */
SDPA_API int sdpa_dev_shard_partial_f32(const float *Qf, int ldq, const float *Kf, int ldk,
                                        const float *Vf, int ldv, float *contrib, int ldo,
                                        float *lmax, float *lsum, int m, int n_local,
                                        int dk, int dv, void *workspace, size_t workspace_bytes,
                                        void *stream);

/* The single-shard call (ABI 6): the stage above followed by merge step 5 with gsum = lsum and the fp64 writeback
 * (attention-mpi.c:358-362, :373) -- result[m x dv] dense fp64 -- with that finish fused into the merge of the in-GPU K/V splits where
 * the launch has splits: one pass over the partial triples instead of three kernels.  contrib / lmax / lsum are scratch of the sizes
 * sdpa_dev_shard_partial_f32 takes (lmax / lsum receive the merged statistics); the rows are bit for bit what
 * sdpa_dev_shard_partial_f32 + sdpa_dev_finish_f64 produce.                                                                         */
SDPA_API int sdpa_dev_shard_attention_f64(const float *Qf, int ldq, const float *Kf, int ldk, const float *Vf, int ldv,
                                          float *contrib, int ldo, float *lmax, float *lsum, double *result, int m,
                                          int n_local, int dk, int dv, void *workspace, size_t workspace_bytes,
                                          void *stream);

/* Merge step 3 (attention-mpi.c:346-351): corr = expf(lmax - gmax);
 * lsum *= corr; contrib row *= corr.                                          */
SDPA_API int sdpa_dev_merge_rescale(float *contrib, int ldo, float *lsum, const float *lmax,
                                    const float *gmax, int m, int dv, void *stream);
/* Merge step 5 (:358-362): inv = gsum==0 ? 0 : 1/gsum; contrib row *= inv.    */
SDPA_API int sdpa_dev_merge_normalise(float *contrib, int ldo, const float *gsum, int m,
                                      int dv, void *stream);
/* Merge steps 2-5 in one pass for hosts that ALL-GATHER the per-shard statistics instead of
 * running the two all-reduces (same algebra): stats[parts][2][m] fp32 with
 * stats[p][0][r] = lmax and stats[p][1][r] = lsum of shard p; `self` is this shard's index.
 * contrib row *= expf(lmax_self - gmax) / gsum, gmax/gsum as attention-mpi.c:342-355.        */
SDPA_API int sdpa_dev_merge_gathered(float *contrib, int ldo, const float *stats, int parts,
                                     int self, int m, int dv, void *stream);
/* Single-shard finish: step 5 with gsum = lsum fused with the fp32->fp64
 * writeback of :373,:396.  result[m x dv] dense fp64.                         */
SDPA_API int sdpa_dev_finish_f64(const float *contrib, int ldo, const float *lsum,
                                 double *result, int m, int dv, void *stream);

/* ---- device level, bf16-input MFMA variant (BASELINE.json config 5) ---------- */
/* Same stage as sdpa_dev_shard_partial_f32 with operands rounded to bf16 (RNE) and
 * fp32 accumulation; tolerance 1e-2*max(1,max|V|).  Operand images:
 *   Qb[m x ld]                     bf16 row-major, ld = sdpa_dev_bf16_ld(dk) (dk padded to
 *                                  64/128/256/512), pad columns zero, holding
 *                                  bf16(Q * log2(e)/sqrtf(dk)): the softmax scale
 *                                  (attention-mpi.c:208) and the change of base for v_exp_f32 are
 *                                  folded into the operand BEFORE its one rounding
 *                                  (sdpa_dev_cvt_d2bf_q) -- the kernels' score chains deliver
 *                                  exp2-domain scores and a softmax weight costs one instruction;
 *   K and V                        as operand IMAGES whose layout belongs to the kernel of the
 *                                  shape's dv (ABI 6).  Write them with sdpa_dev_cvt_d2bf_k /
 *                                  sdpa_dev_cvt_d2bf_v (or, on the host, sdpa_host_cvt_rows kind 1 / 2
 *                                  and sdpa_host_cvt_vt); sizes: K ldn x ld elements, V dvp x ldn
 *                                  elements, ldn = sdpa_dev_bf16_ldn(n_local) (n_local padded to
 *                                  32), dvp = sdpa_dev_bf16_dvp(dv) -- ALLOCATE the pad rows of K.
 *     dv <= 256 (sdpa_dev_bf16_tiled(dv) == 0), "row images":
 *       Kb[n_local x ld]           row-major, pad columns zero;
 *       Vt[dvp x ldvt]             V TRANSPOSED with the keys of a row permuted inside 16-key
 *                                  groups: Vt[c*ldvt + sdpa_dev_bf16_kvpos(j)] = V[j][c], kvpos(j) =
 *                                  j with bits 2 and 3 swapped (group order 0-3, 8-11, 4-7, 12-15:
 *                                  the eight keys one MFMA lane multiplies are then 16 contiguous
 *                                  bytes); ldvt = ldn, pads zero.
 *     dv > 256 (== 1), "tiled images" (round 6): every 32-key tile is one contiguous block in the
 *     byte order of the kernel's LDS buffers, so that its LDS-DMA pieces are lane-linear at both ends
 *     (one scalar base per tile, immediates per piece -- DESIGN.md 4.2):
 *       Kb[ldn x ld]               the rows above with 16-byte chunk c of row r stored at chunk
 *                                  position c ^ (r & min(15, ld/8 - 1)); rows [n_local, ldn) are
 *                                  read and masked (sdpa_dev_cvt_d2bf_k zeroes them);
 *       Vt[ldn/32][dvp/512][512][32]  per key tile and 512-column chunk a block of 512 columns x
 *                                  32 key positions (kvpos order), a column's 16-byte chunk q stored
 *                                  at q ^ ((column >> 2) & 3); zero for keys >= n_local and
 *                                  columns >= dv.  ldvt is not used (pass ldn).
 * sdpa_dev_cvt_d2bf (plain rows) and sdpa_dev_cvt_d2bf_t (the V image by its arguments: tiled when
 * cols > 256) remain; a K image for dv > 256 must come from sdpa_dev_cvt_d2bf_k.
 * dk <= 512, dv <= 1024.  contrib/lsum are those of the scores so computed,
 * relative to lmax (natural-log units) as in the fp32 variant -- but lmax is here a REFERENCE
 * EXPONENT, not necessarily the row max: the fixed-reference kernels (dk, dv <= 256, and dv > 256)
 * return the power of two that puts lsum in [1, 2), so lmax + ln(lsum) is the row's log-sum-exp
 * and max_j s_ij <= lmax + ln 2 <= max_j s_ij + ln(2 n_local).  Every merge of this library (and
 * attention-mpi.c:340-380) is invariant under that choice.                                   */
SDPA_API int  sdpa_dev_bf16_ld(int dk);
SDPA_API int  sdpa_dev_bf16_dvp(int dv);
SDPA_API long sdpa_dev_bf16_ldn(long n_local);
SDPA_API long sdpa_dev_bf16_kvpos(long j);
SDPA_API int  sdpa_dev_cvt_d2bf(const double *src, void *dst, long rows, int cols, int ld,
                                void *stream);
SDPA_API int  sdpa_dev_cvt_d2bf_q(const double *src, void *dst, long rows, int dk, int ld,
                                  void *stream);
SDPA_API int  sdpa_dev_cvt_d2bf_t(const double *src, void *dst, long rows, int cols, int cols_pad,
                                  long ldt, void *stream);
/* ABI 6: the images of a (dk, dv) shape, in the layout its kernel reads.  K: `rows` rows of fp64 [rows x dk] into
 * dst[ldn(rows) x ld(dk)] (pad rows zeroed for tiled images).  V: fp64 [rows x dv] into dst[dvp(dv) x ldn(rows)] elements.  */
SDPA_API int  sdpa_dev_bf16_tiled(int dv);
SDPA_API int  sdpa_dev_cvt_d2bf_k(const double *src, void *dst, long rows, int dk, int dv, void *stream);
SDPA_API int  sdpa_dev_cvt_d2bf_v(const double *src, void *dst, long rows, int dv, void *stream);
SDPA_API int    sdpa_dev_kv_splits_bf16(int m, int n_local, int dk, int dv);
SDPA_API size_t sdpa_dev_workspace_bytes_bf16(int m, int n_local, int dk, int dv);
SDPA_API int  sdpa_dev_shard_partial_bf16(const void *Qb, int ldq, const void *Kb, int ldk,
                                          const void *Vt, long ldvt, float *contrib, int ldo,
                                          float *lmax, float *lsum, int m, int n_local, int dk,
                                          int dv, void *workspace, size_t workspace_bytes,
                                          void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SDPA_HIP_H */
