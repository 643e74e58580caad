// sdpa_internal.h -- launchers shared between the kernel translation units and
// the C-ABI layer.  Not part of the public interface (include/sdpa_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// The LDS-DMA asm statements write M0 (the DMA's LDS destination) without saving it.  Declaring
// the clobber tells hipcc so: it cannot then keep a value of its own in M0 across such a statement
// (LDS-direct, movrel, sendmsg users would otherwise silently read the DMA's address).
// -DSDPA_M0_CLOBBER= (empty) builds the undeclared form for A/B timing (tools/build_variant.sh).
#ifndef SDPA_M0_CLOBBER
#define SDPA_M0_CLOBBER , "m0"
#endif

// -DSDPA_DMA_ASSERT (tools/build_variant.sh, never shipped): every global source address of an LDS-DMA piece and of
// a clamped fragment load is checked against the operand image it must lie in; a violation bumps a per-translation-
// unit device counter that sdpa_debug_dma_audit() reads (tests/conftest.py fails the session on a non-zero count).
// The fp32 pipelined kernel's older audit poisons its row sums instead (sdpa_fwd_f32.hip).
#ifdef SDPA_DMA_ASSERT
#define SDPA_AUDIT_COUNTER(name) static __device__ unsigned long long name[2] = {0ull, 0ull};
#define SDPA_AUDIT(counter, src, bytes, lo, hi)                                                                  \
    do {                                                                                                         \
        const char *s_ = reinterpret_cast<const char *>(src);                                                    \
        if (s_ < reinterpret_cast<const char *>(lo) || s_ + (bytes) > reinterpret_cast<const char *>(hi) ||      \
            (reinterpret_cast<unsigned long long>(s_) & 3ull) != 0)                                              \
            atomicAdd(&counter[0], 1ull);                                                                        \
    } while (0)
#define SDPA_AUDIT_LAUNCH(counter) do { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&counter[1], 1ull); } while (0)
// the pointer itself, checked on the way (expands to the bare pointer in the shipped build: same code as without it)
__device__ inline const char *sdpa_audited_ptr(unsigned long long *counter, const char *p, unsigned bytes, const void *lo,
                                               const void *hi) {
    if (p < reinterpret_cast<const char *>(lo) || p + bytes > reinterpret_cast<const char *>(hi) ||
        (reinterpret_cast<unsigned long long>(p) & 3ull) != 0)
        atomicAdd(&counter[0], 1ull);
    return p;
}
#define SDPA_AUDITED_PTR(counter, p, bytes, lo, hi) sdpa_audited_ptr(counter, p, bytes, lo, hi)
#else
#define SDPA_AUDITED_PTR(counter, p, bytes, lo, hi) (p)
#define SDPA_AUDIT_COUNTER(name)
#define SDPA_AUDIT(counter, src, bytes, lo, hi) ((void)0)
#define SDPA_AUDIT_LAUNCH(counter) ((void)0)
#endif

namespace sdpa {

// audit builds: [0] = violations, [1] = audited launches of the translation unit's kernels (0, 0 otherwise)
void dma_audit_read_bf16(unsigned long long out[2]);
void dma_audit_read_dksplit(unsigned long long out[2]);

constexpr int kQRowsPerBlock = 128;   // 4 waves x 32 query rows
constexpr int kKvTile        = 32;    // K/V rows per LDS tile
constexpr int kMaxFastDim    = 128;   // dk, dv <= 128: one dv chunk, two workgroups per CU
constexpr int kMaxMfmaDk     = 256;   // dk <= 256 takes an MFMA kernel (any dv, 128-column chunks)
// Leading dimension the fp32 operand images and the contrib rows are best given: head dims in (32, 256]
// padded (with zero columns) to 64 / 128 / 256, so that they take the LDS-DMA pipelined kernel whatever
// the dims are -- it does the padded MFMA work the any-shape kernels do as well, at its own rate.
// Beyond 256 (the dk-split kernels): a whole number of the lane runs its dv slice reads -- 3 / 4 / 6 / 8
// consecutive V columns for the 96 / 128 / 192 / 256-wide slices of dv <= 384 / 512 / 768 / 1024+ -- and of 4,
// so that no run straddles the row end: multiples of 12 / 4 / 12 / 8 (384, 512, 768, 1024 stay as they are).
inline int dense_ld(int d) {
    if (d <= 256) return d <= 32 ? (d + 3) / 4 * 4 : d <= 64 ? 64 : d <= 128 ? 128 : 256;
    const int q = d <= 384 ? 12 : d <= 512 ? 4 : d <= 768 ? 12 : 8;
    return (d + q - 1) / q * q;
}
constexpr int kMaxDkSplit    = 1024;  // 256 < dk <= 1024: the dk-split MFMA kernel (waves split dk and dv)

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
    long ws_rows;                  // rows between two slots of the ws arrays (0 = m): lets a launch over a
                                   // sub-range of rows write into the slots of a larger batch
    int defer_merge;               // 1 = leave the kv_splits partial triples in the ws slots (the host
                                   // pipeline merges the slots of all its K/V chunks in one pass)
    unsigned long long *tickets;   // [query blocks of 128 rows] arrival words of the IN-KERNEL split merge (pipelined
                                   // fp32 kernels with kv_splits > 1 and no defer_merge): the last workgroup of a
                                   // query block to arrive merges its splits.  nullptr = separate merge pass
    unsigned long long ticket_tag; // this launch's generation (set by the launcher): a word counts only when its
                                   // upper 56 bits equal it, so nothing has to be cleared between launches
    int tune;                      // -DSDPA_ABLATIONS builds only ($SDPA_DEBUG tune): 4 = register-staged kernel
                                   // instead of the pipelined one, 16/32/64 = timing-only ablations
    int cus;                       // compute units this launch may fill (stream-K grid); 0 = what the stream was
                                   // registered with (register_stream_cus), the whole chip otherwise
};

// The persistent chunk-following launch (fused_pipelined_stream_kernel, sdpa_fwd_f32_pipelined.inc): where its ready
// words live and what they announce.  Passed by value beside PartialArgs.
constexpr int kStreamMaxChunks = 16;      // ready words [0, 16): K/V chunks; [16, 16 + kStreamMaxPieces): Q row pieces
constexpr int kStreamMaxPieces = 8;
// A ready word is raised by the COPY ENGINE, and on this runtime only copies of some size are the copy engine's: a 4-byte
// host-to-device copy (and hipStreamWriteValue32) is carried out by a shader blit, which cannot become resident while
// the persistent launch owns every register of every SIMD -- it lands when the launch ends (tools/probes/
// stream_flag_probe2.hip, profiles/r05/stream_flag_probe2.log: never seen within 300 ms; a 64 KiB copy: seen at once).
// So every ready word is the first word of its own 64 KiB block, and raising it copies 64 KiB of generation words.
constexpr int kStreamFlagStride = 16384;  // words between two ready words (64 KiB)
struct StreamArgs {
    const unsigned *flags;                // device words, kStreamFlagStride apart; a word counts when it equals `gen`
    unsigned gen;                         // this launch's generation (never 0; nothing is cleared between launches)
    int n_chunks;
    int chunk_end[kStreamMaxChunks];      // tiles of a split's K/V range that are on the device once chunk c has landed
                                          // (strictly increasing; the last one covers the longest split)
    int q_piece_blocks;                   // query blocks per Q row piece (0 = Q is resident before the launch)
    int q_piece0;                         // ... and the piece the launch's first query block belongs to (a launch over the second half of a batch)
    unsigned long long timeout_ticks;     // wall_clock64 ticks (100 MHz) one wait may last
    int *status;                          // device-visible word, set to 1 + the word's index when a wait timed out (results are garbage then)
    unsigned *abort;                      // device word: == gen once a wait of this launch has timed out (the others stop waiting)
};
// dims the stream form exists for: dense images 64 / 128 wide on both sides (every BASELINE fp32 shape with d <= 128)
bool stream_launch_supported(int dk, int dv);
// classic grid ceil(m/128) x a.kv_splits (equal K/V ranges), ONE launch; + the split merge when kv_splits > 1 and
// !defer_merge.  The triples are bit for bit those of launch_shard_partial() with the same kv_splits on resident inputs.
hipError_t launch_shard_partial_streamed(const PartialArgs &a, const StreamArgs &st, hipStream_t s);

// bf16 variant: Q row-major bf16 (ld = dk padded to 64/128/256/512, pad columns zero), holding bf16(Q log2e / sqrt(dk)).
// K and V come as operand IMAGES whose layout belongs to the kernel of the shape's dv:
//   row images (dv <= 256: duo / pipe kernels)
//     K  [n_local x ldk] row-major, pad columns zero;
//     Vt [dv_pad x ldvt] = V transposed (ldvt = n_local padded to 32, pads zero), key j of a row stored at bf16_kvpos(j).
//   tiled images (dv > 256: the tandem kernel and its redo pass; round 6) -- every 32-key tile is one contiguous block in
//   exactly the byte order of the kernel's LDS buffer, so that its LDS-DMA pieces are lane-linear at both ends:
//     K  [pad_n(n_local) x ldk]: row-major rows, 16-byte chunk c of row r stored at chunk position c ^ (r & bf16_k_swz)
//        (a tile = 32 rows = 32 * ldk contiguous elements); rows [n_local, pad_n) are read (and masked): any finite-or-not
//        content, but ALLOCATED;
//     Vt [pad_n / 32 tiles][dv_pad / 512 chunks][512 columns][32 key positions]: a column's 64 bytes hold the tile's keys in
//        bf16_kvpos order, 16-byte chunk q stored at q ^ ((column >> 2) & 3); zero for keys >= n_local and columns >= dv.
//     The image of keys [k0, k0 + cnt) (k0 a multiple of 32) starts bf16_vt_key_offset(k0, dv) elements into the Vt image
//     and k0 * ldk elements into the K image: a launch over a sub-range of keys takes those pointers.
struct Bf16Args {
    const unsigned short *Q;  int ldq;
    const unsigned short *K;  int ldk;
    const unsigned short *Vt; long ldvt;       // (ldvt: row images only)
    float *contrib;  int ldo;
    float *lmax;
    float *lsum;
    int m, n_local, dk, dv;
    int kv_splits;
    float *ws_contrib;  int ws_ld;
    float *ws_lmax;
    float *ws_lsum;
    long ws_rows;                  // as PartialArgs::ws_rows
    int defer_merge;               // as PartialArgs::defer_merge
    int *redo;                     // [kv_splits x q blocks] flags, dv > 256 only (bf16_carve_workspace)
    int redo_gen;                  // a flag counts when it equals this launch's generation (set by the
                                   // launcher): nothing has to be cleared, stale or uninitialised
                                   // values at worst cause a harmless extra redo
    int tiled;                     // set by the launcher: bf16_tiled(dv) -- the layout of K and Vt (kernels that serve both read it)
};

int  bf16_pad_dk(int dk);
int  bf16_chunk_dv(int dv);
int  bf16_pad_dv(int dv);
long bf16_pad_n(long n);
// position of key j inside a Vt row: bits 2 and 3 of j swapped (an involution) -- each 16-key
// group is stored 0-3, 8-11, 4-7, 12-15 so one MFMA lane's eight keys are 16 contiguous bytes
__host__ __device__ inline constexpr long bf16_kvpos(long j) {
    return (j & ~12L) | ((j & 4) << 1) | ((j & 8) >> 1);
}
// which image layout a shape's kernels read, and where things sit in it (comment above Bf16Args)
inline bool bf16_tiled(int dv) { return dv > 256; }
inline int  bf16_k_swz(int dk, int dv) {
    if (!bf16_tiled(dv)) return 0;
    const int kch = (dk <= 64 ? 64 : dk <= 128 ? 128 : dk <= 256 ? 256 : 512) / 8;    // 16-byte chunks of a K row
    return kch >= 16 ? 15 : kch - 1;
}
inline size_t bf16_vt_key_offset(long k0, int dv) {       // elements from the image's start to the block of key k0 (multiple of 32)
    if (!bf16_tiled(dv)) return (size_t)k0;
    return (size_t)k0 * (size_t)((dv + 511) / 512 * 512);
}
bool bf16_needs_redo(int dk, int dv);  // the shape's kernel flags blocks for a second pass (workspace holds the flags)
int  pick_kv_splits_bf16(int m, int n_local, int dk, int dv);
size_t bf16_workspace_bytes(int m, int n_local, int dk, int dv);
void bf16_carve_workspace(Bf16Args &a, void *ws, int ws_ld);   // needs a.m, a.dv, a.kv_splits
hipError_t launch_shard_partial_bf16(const Bf16Args &a, hipStream_t s);
// the persistent, input-following form of the same launch (StreamArgs as for the fp32 one; tile = kKvTile keys of a split's
// range): dims with bf16_stream_launch_supported() only.  Triples bit for bit those of launch_shard_partial_bf16().
bool bf16_stream_launch_supported(int dk, int dv);
hipError_t launch_shard_partial_bf16_streamed(const Bf16Args &a, const StreamArgs &st, hipStream_t s);
hipError_t launch_cvt_d2bf(const double *src, unsigned short *dst, long rows, int cols, int ld, hipStream_t s);
// the K image of a (dk, dv) shape at `dst` (the image's row of the first key, a multiple of 32 keys in): rows [rows, rows_pad) zero
hipError_t launch_cvt_d2bf_k(const double *src, unsigned short *dst, long rows, long rows_pad, int dk, int dv, hipStream_t s);
hipError_t launch_cvt_d2bf_q(const double *src, unsigned short *dst, long rows, int dk, int ld, hipStream_t s);
hipError_t launch_cvt_d2bf_t(const double *src, unsigned short *dst, long rows, int cols, int cols_pad,
                             long ldt, hipStream_t s);
// the same for `rows` keys that land at dst (row images: a column offset into a larger Vt image of row stride ldt, zero-fills
// key positions [rows, rows_pad) only; tiled images -- chosen by cols > 256 --: dst = image + bf16_vt_key_offset(first key),
// whole tiles are written)
hipError_t launch_cvt_d2bf_t_part(const double *src, unsigned short *dst, long rows, long rows_pad, int cols,
                                  int cols_pad, long ldt, hipStream_t s);
// the same from rows that are bf16 already (dense, row stride = cols): $SDPA_HOST_CVT, where the host rounds
hipError_t launch_cvt_bf_t_part(const unsigned short *src, unsigned short *dst, long rows, long rows_pad, int cols,
                                int cols_pad, long ldt, hipStream_t s);
hipError_t launch_split_merge(const PartialArgs &a, hipStream_t s);
// ... with the single-shard finish fused in (round 6): the merged rows leave normalised and dense [m x dv], as fp64 and / or fp32
// (split_merge_finish_kernel, sdpa_fwd_f32.hip); a.lmax / a.lsum (optional) receive the merged statistics, a.contrib is not written
struct FinishTarget {
    double *out64;
    float *out32;
};
hipError_t launch_split_merge_finish(const PartialArgs &a, const FinishTarget &f, hipStream_t s);
// The runtime loads a translation unit's device code object when one of its kernels is first used -- an upload that needs the GPU and
// so WAITS for a resident persistent launch: the split merge enqueued right behind the first streamed bf16 launch of a process blocked
// its enqueuing thread for the launch's whole timeout, in FRONT of the copies the launch was waiting for (round 5, call 22).  The engine
// loads every unit's code object when it creates a rank (current device).
hipError_t preload_kernels_f32();
hipError_t preload_kernels_dksplit();
hipError_t preload_kernels_bf16();
hipError_t preload_kernels_aux();
hipError_t preload_kernels_coll();

// In-GPU K/V splits against TAIL QUANTISATION.  `blocks * s` workgroups run in ceil(blocks * s / slots) rounds of
// `slots` resident workgroups; once there is more than one round, a thinly filled last round idles most of the
// chip (m = 40000 at d = 128: 313 query blocks x 2 splits = 626 workgroups = 1.22 rounds: 61 %).  From the count
// `s0` that fills the chip once, look further for a fuller last round; every extra split costs one more slab
// written and read by the merge, so the choice minimises  kernel time / efficiency(s) + s x slab time.
// Shapes whose workgroups fit one round keep s0 (all BASELINE shapes do: 64, 256, 1024 query blocks).
inline int splits_for_full_rounds(long blocks, int slots, int s0, int cap, double kernel_s, double slab_s) {
    if (blocks * s0 <= slots || kernel_s <= 0.0) return s0;
    auto cost = [&](int sp) {
        const long w = blocks * sp, rounds = (w + slots - 1) / slots;
        return kernel_s * (double)(rounds * slots) / (double)w + sp * slab_s;
    };
    int best = s0;
    double bc = cost(s0);
    const int hi = cap < 64 ? cap : 64;
    for (int sp = s0 + 1; sp <= hi && sp <= s0 + 14; ++sp) {
        const double cst = cost(sp);
        if (cst < bc * 0.98) {           // a clear win only: near-ties keep the smaller count
            best = sp;
            bc = cst;
        }
    }
    return best;
}

// ---- how one fp32 fused launch distributes its work (round 4) ------------------------------------------------
// classic:   ceil(m/128) query blocks x `splits` equal K/V ranges = one workgroup each; fast when that count is a
//            whole number of rounds of the stream's resident workgroup slots (every BASELINE shape on a whole chip)
// stream-K:  `workers` = the resident slots; every workgroup walks `run` consecutive tile steps of the launch's
//            (query block, K/V tile) space and leaves one partial triple per query block it touches; `splits` =
//            the most pieces any query block is cut into = slabs of split scratch the launch needs.  Chosen when
//            the classic grid would leave slots idle: a CU-masked stream, an odd m, a short shard.
// `cus` = compute units the launch's stream may use (0 = a whole MI355X, kChipCus).
constexpr int kChipCus = 256;
struct F32Plan {
    int splits;     // slabs of split scratch (1 = the launch writes the result rows itself)
    int streamk;    // 1 = stream-K distribution (pipelined 64/128/256-wide kernels only)
    int workers;    // stream-K: workgroups
    int run;        // stream-K: tile steps per workgroup
};
F32Plan plan_f32_launch(int m, int n_local, int dk, int dv, int cus);
int  pick_kv_splits(int m, int n_local, int dk, int dv, int cus = 0);
// scratch that covers the launch on ANY stream (whole chip or CU-masked: the slab counts differ)
size_t workspace_bytes(int m, int n_local, int dk, int dv);
size_t workspace_bytes_for(int m, int dv, int splits);

// Compute units a stream's kernels may use: what create_masked_stream() registered for it, otherwise the
// current device's count (a stream this library did not create).  Thread-safe.
int  stream_cus(hipStream_t s);
void register_stream_cus(hipStream_t s, int cus);
void forget_stream_cus(hipStream_t s);

// What the calling thread's LAST fused launch was: the kernel as rocprofv3 prints it (name + template arguments), its
// grid and its work distribution.  Written by every fused launcher (a handful of stores), read by
// sdpa_dev_last_launch() and the host pipeline (sdpa_timing.last_kernel): bench.py's roofline.kernel is what
// LAUNCHED, not what a heuristic in the bench expects to launch.
struct LaunchNote {
    const char *kernel;   // e.g. "fused_pipelined_kernel" (a string literal), nullptr = none yet
    int targ[5];          // template arguments in declaration order
    int ntarg;
    int grid;             // workgroups
    int splits;           // slabs of split scratch (classic: equal K/V splits; stream-K: most pieces per query block)
    int streamk;          // 1 = stream-K distribution
    int rows, keys;       // launch shape
};
void note_launch(const char *kernel, int ntarg, int t0, int t1, int t2, int t3, int t4, int grid, int splits, int streamk,
                 int rows, int keys);
const LaunchNote &last_launch_note();
void set_launch_note(const LaunchNote &n);   // (the bf16 launchers put the main kernel back behind its redo pass)
// "sdpa::name<a,b,...>" into buf (always terminated); returns the length it would need
int format_launch_kernel(const LaunchNote &n, char *buf, size_t len);

// Launch-path knobs.  The launchers run on the hosts' enqueue threads, and glibc's environment is not safe to
// read while another thread may setenv(): the knobs are read ONCE (first use) into an immutable snapshot;
// sdpa_reload_env() (and every host-level entry point, on the calling thread, before any worker thread
// runs) takes a new snapshot.
struct LaunchKnobs {
    int split_merge_kernel;   // $SDPA_DEBUG split_merge=kernel: the fp32 pipelined kernel merges its K/V splits itself
    int streamk;              // $SDPA_DEBUG streamk: 0 = never, 1 = whenever eligible, unset/auto = -1: by the cost model
};
const LaunchKnobs &launch_knobs();
void reload_launch_knobs();
// Point a.ws_* (and a.tickets) into a scratch area of workspace_bytes(a.m, ...) bytes: kv_splits slabs of
// a.m rows x ws_ld floats, the two statistics arrays, the arrival words.  Needs a.m, a.kv_splits.
void carve_workspace(PartialArgs &a, void *ws, int ws_ld);

// the dk-split kernels (sdpa_fwd_f32_dksplit.hip): launcher and the grid arithmetic pick_kv_splits needs
hipError_t launch_dksplit(const PartialArgs &a, hipStream_t s);
int dksplit_chunks(int dk, int dv);
int dksplit_rows(int dk);

// Enqueue the fused kernel (+ the split merge when kv_splits > 1).
hipError_t launch_shard_partial(const PartialArgs &a, hipStream_t s);

hipError_t launch_cvt_d2f(const double *src, float *dst, long rows, int cols, int ld, hipStream_t s);
// operand-like patterns (fp32 in [-mag, mag), or pairs of bf16) for sdpa_prepare()'s warm-up launches; bytes rounded down to words
hipError_t launch_fill_pattern(void *dst, size_t bytes, int bf16, float mag, hipStream_t s);
// up to three fp64 -> fp32 images in one launch (each exactly launch_cvt_d2f's)
hipError_t launch_cvt_d2f_batch(int count, const double *const *src, float *const *dst, const long *rows, const int *cols, const int *ld,
                                hipStream_t s);
hipError_t launch_cvt_f2d(const float *src, int ld, double *dst, long rows, int cols, hipStream_t s);
hipError_t launch_merge_rescale(float *contrib, int ldo, float *lsum, const float *lmax,
                                const float *gmax, int m, int dv, hipStream_t s);
hipError_t launch_merge_normalise(float *contrib, int ldo, const float *gsum, int m, int dv,
                                  hipStream_t s);
hipError_t launch_merge_gathered(float *contrib, int ldo, const float *stats, int parts, int self,
                                 int m, int dv, hipStream_t s);
hipError_t launch_finish_f64(const float *contrib, int ldo, const float *lsum, double *result,
                             int m, int dv, hipStream_t s);
// the same rows left in fp32, dense (dv floats a row); lsum == nullptr: repack only
hipError_t launch_finish_f32(const float *contrib, int ldo, const float *lsum, float *out, int m, int dv,
                             hipStream_t s);

}  // namespace sdpa
