/* sdpa_debug.h -- $SDPA_DEBUG="name=value,name=value": ONE environment variable for every test, tuning and experiment knob of the
 * library and its hosts (round 6, VERDICT r5 item 8: 43 separate SDPA_* variables before).  The documented runtime knobs -- the ones a
 * user of the drop-in may want -- keep their own variables and are listed in include/sdpa_hip.h; everything here is for the test-suite,
 * tools/ and A/B runs, may change between rounds, and is read where the old variable was read (same thread, same moment).
 * Plain C (the gcc-built hosts include it too).  Names (default):
 *   planner     kv_chunk_min (4096), kv_chunk_max (per problem), row_pieces (4), piece_min_rows (4096), stream_chunk_min (= kv_chunk_min),
 *               stream_entry_min (2048), stream_interleave (1), stream_interleave_min (2048), stream_interleave_max (per problem), stream_pair (1), stream_q_with_group0 (1), egress_in_order (1), stream_probe_ms (200), two_wave (0), stream_drop_word (0 = none), enqueue_threads (1),
 *               progressive_pin (1), pin_probe (1), host_probe (0), host_cores (from the cgroup quota), reserve_by_mask (0),
 *               force_collectives (0: 1 = the merge's collectives run on one rank too, real RCCL on a one-rank communicator),
 *               prepare_zero (0: 1 = sdpa_prepare() warms the clock on zeroed operands, as rounds 3-5)
 *   converters  host_cvt_item_kb (64), host_cvt_nt (1), host_cvt_pin (0), host_cvt_trace (0)
 *   launchers   split_merge (separate | kernel), streamk (auto | 0 | 1), bf16_duo (1), tune (0; -DSDPA_ABLATIONS builds only)
 *   CLI hosts   pinned_io (1), time_init (0)                                                                                        */
#ifndef SDPA_DEBUG_H
#define SDPA_DEBUG_H
#include <stdlib.h>
#include <string.h>

/* -> the value of `name` inside $SDPA_DEBUG (NOT terminated: it ends at ',' or at the string's end), or NULL */
static inline const char *sdpa_debug_find(const char *name) {
    const char *v = getenv("SDPA_DEBUG");
    if (!v || !*v) return NULL;
    const size_t len = strlen(name);
    const char *p = v;
    while (*p) {
        while (*p == ',' || *p == ' ') ++p;
        if (strncmp(p, name, len) == 0 && p[len] == '=') return p + len + 1;
        while (*p && *p != ',') ++p;
    }
    return NULL;
}
/* the value as an integer (atoi stops at the ','), `dflt` when the name is absent or its value empty */
static inline int sdpa_debug_int(const char *name, int dflt) {
    const char *v = sdpa_debug_find(name);
    return (v && *v && *v != ',') ? atoi(v) : dflt;
}
/* ... where only positive values count (the planner's sizes) */
static inline int sdpa_debug_pos(const char *name, int dflt) {
    const int x = sdpa_debug_int(name, 0);
    return x > 0 ? x : dflt;
}
/* the value equals `word` */
static inline int sdpa_debug_is(const char *name, const char *word) {
    const char *v = sdpa_debug_find(name);
    if (!v) return 0;
    const size_t len = strlen(word);
    return strncmp(v, word, len) == 0 && (v[len] == 0 || v[len] == ',');
}
#endif
