/*
 * attention-hip.c -- plain-C host program with the reference's CLI contract, whose
 * attention() body is the MI355X engine behind include/sdpa_hip.h.
 *
 * Contract kept (paths relative to the reference tree):
 *   usage / exit codes            attention.c:165-168
 *   input file format             attention.c:92-121  (4 x int32 m,n,dk,dv; Q,K,V fp64)
 *   answer block + 0.02 check     attention.c:123-162 (incl. the template's NaN probe of
 *                                 column 1 only, :150 -- stdout must not differ from the
 *                                 reference for the same result array)
 *   stdout                        attention.c:184-189 ("Correct!\nElapsed time: %.2lf us\n"
 *                                 or "Wrong!\n"); diagnostics go to stderr only
 *   timed region                  attention.c:179-182 (the attention() call)
 *   boundary                      attention.c:20-21   void attention(double*,double*,double*,
 *                                                     double*,int,int,int,int)
 *
 * Built by gcc; sees nothing but the C header.  Environment:
 *   SDPA_GPUS=N        GPUs to shard K/V over (default: all visible)
 *   SDPA_TIME_INIT=1   create and size the engine inside the timed region (default: before
 *                      it -- sdpa_init + sdpa_prepare -- the way the reference sets up MPI and
 *                      its transport outside the timer, attention-mpi.c:10-17, :504)
 *   SDPA_VERBOSE=1     stage breakdown and a strict parity report on stderr
 *   SDPA_PINNED_IO=0   read the matrices into malloc'd memory (default: page-locked memory from
 *                      sdpa_host_alloc when the engine is created before the read -- the file
 *                      format and the reader's error behaviour stay attention.c:84-121)
 */
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "sdpa_hip.h"

static void die_if(int code, const char *what)
{
    if (code == SDPA_OK) return;
    fprintf(stderr, "attention-hip: %s: %s\n", what, sdpa_strerror(code));
    exit(1);
}

/* ---- the drop-in boundary ------------------------------------------------ */
void attention(double *Q, double *K, double *V, double *result,
               int m, int n, int dk, int dv)
{
    die_if(sdpa_attention_f64(Q, K, V, result, m, n, dk, dv, SDPA_F_DEFAULT),
           "sdpa_attention_f64");
}

/* ---- file handling ------------------------------------------------------- */
struct problem {
    int32_t dim[4];            /* m, n, dk, dv */
    double *q, *k, *v;
};

static void bad_data(void)
{
    fprintf(stderr, "Invalid testing data.\n");
    exit(1);
}

/* matrices live in page-locked memory when the engine can give it: no
 * registration pass inside the timed call, full-rate H2D from the first touch */
static bool use_pinned = false;

static struct { double *p; bool pinned; } host_bufs[8];
static int n_host_bufs = 0;

static double *host_doubles(size_t count)
{
    double *buf = NULL;
    bool pinned = false;
    if (use_pinned) {
        buf = (double *)sdpa_host_alloc(count * sizeof(double));
        pinned = buf != NULL;
    }
    if (!buf) buf = (double *)malloc(count * sizeof(double));
    if (buf && n_host_bufs < 8) {
        host_bufs[n_host_bufs].p = buf;
        host_bufs[n_host_bufs].pinned = pinned;
        ++n_host_bufs;
    }
    return buf;
}

static void release_host_bufs(void)
{
    for (int i = 0; i < n_host_bufs; ++i) {
        if (host_bufs[i].pinned) sdpa_host_free(host_bufs[i].p);
        else free(host_bufs[i].p);
    }
    n_host_bufs = 0;
}

static double *slurp(FILE *f, size_t count)
{
    double *buf = host_doubles(count);
    if (!buf || fread(buf, sizeof(double), count, f) != count) bad_data();
    return buf;
}

/* What load_problem() would say about this file, decided from its header and size alone --
 * so that bad input is reported before any device is touched (same messages, same exit code). */
static void precheck_file(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "Cannot open file: %s\n", path);
        exit(1);
    }
    int32_t d[4];
    for (int i = 0; i < 4; ++i)
        if (fread(&d[i], sizeof(int32_t), 1, f) != 1) bad_data();
    const double need = 16.0 + 8.0 * ((double)d[0] * d[2] + (double)d[1] * d[2] + (double)d[1] * d[3]);
    if (fseek(f, 0, SEEK_END) != 0 || (double)ftell(f) < need) bad_data();
    fclose(f);
}

static void load_problem(const char *path, struct problem *p)
{
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "Cannot open file: %s\n", path);
        exit(1);
    }
    for (int i = 0; i < 4; ++i)
        if (fread(&p->dim[i], sizeof(int32_t), 1, f) != 1) bad_data();
    const size_t m = (size_t)p->dim[0], n = (size_t)p->dim[1];
    const size_t dk = (size_t)p->dim[2], dv = (size_t)p->dim[3];
    p->q = slurp(f, m * dk);
    p->k = slurp(f, n * dk);
    p->v = slurp(f, n * dv);
    fclose(f);
}

/* Compare against the answer block appended to the input file.  Returns the
 * reference's verdict; *worst receives the largest |difference| seen up to the
 * point the reference would have stopped (all rows when it passes). */
static bool check_answer(const char *path, const double *result, double *worst, long *nonfinite)
{
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "Cannot open answer file: %s\n", path);
        return false;
    }
    int32_t d[4];
    for (int i = 0; i < 4; ++i)
        if (fread(&d[i], sizeof(int32_t), 1, f) != 1) bad_data();
    const int m = d[0], n = d[1], dk = d[2], dv = d[3];
    /* the template computes this offset in int (attention.c:139); Q+K+V < 2 GiB */
    const long skip = 16L + 8L * ((long)m * dk + (long)n * dk + (long)n * dv);
    fseek(f, skip, SEEK_SET);

    const double tol = 0.02;
    double *want = (double *)malloc(sizeof(double) * (size_t)dv);
    bool ok = true;
    *worst = 0.0;
    *nonfinite = 0;
    for (int i = 0; i < m && ok; ++i) {
        const double *got = result + (size_t)i * dv;
        if (fread(want, sizeof(double), (size_t)dv, f) != (size_t)dv) {
            /* the template ignores a short answer block and compares stale data;
             * a missing answer cannot be "Correct!" here */
            ok = false;
            fprintf(stderr, "attention-hip: answer block truncated at row %d\n", i);
            break;
        }
        /* the template probes only column 1 of the row for NaN (attention.c:150) */
        const bool nan_probe = dv > 1 ? isnan(got[1]) : false;
        for (int j = 0; j < dv; ++j) {
            const double gap = fabs(got[j] - want[j]);
            if (!isfinite(got[j])) ++*nonfinite;
            if (gap > *worst) *worst = gap;
            if (nan_probe || gap > tol) {
                printf("Expect result[%d][%d] to be %lf, but it is %lf\n", i, j, want[j], got[j]);
                ok = false;
                break;
            }
        }
    }
    free(want);
    fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "Usage: %s <testing data>\n", argv[0]);
        return 1;
    }
    const bool verbose = getenv("SDPA_VERBOSE") != NULL;
    const bool time_init = getenv("SDPA_TIME_INIT") != NULL;
    const char *gpus = getenv("SDPA_GPUS");

    /* the engine comes up before the read so that the reader can ask it for page-locked memory;
     * a missing or truncated input file is still reported first, with the reader's own messages
     * (attention.c:102-114) */
    if (!time_init) {
        precheck_file(argv[1]);
        die_if(sdpa_init(gpus ? atoi(gpus) : 0), "sdpa_init");
        const char *pin = getenv("SDPA_PINNED_IO");
        use_pinned = !(pin && pin[0] == '0');
    }

    struct problem p;
    load_problem(argv[1], &p);
    const int m = p.dim[0], n = p.dim[1], dk = p.dim[2], dv = p.dim[3];
    double *result = host_doubles((size_t)m * (size_t)dv);
    if (!result) {
        fprintf(stderr, "attention-hip: out of memory\n");
        return 1;
    }

    if (!time_init) die_if(sdpa_prepare(m, n, dk, dv, SDPA_F_DEFAULT), "sdpa_prepare");

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    attention(p.q, p.k, p.v, result, m, n, dk, dv);
    clock_gettime(CLOCK_MONOTONIC, &t1);

    double worst = 0.0;
    long nonfinite = 0;
    if (check_answer(argv[1], result, &worst, &nonfinite)) {
        const double us = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) / 1e3;
        printf("Correct!\nElapsed time: %.2lf us\n", us);
    } else {
        puts("Wrong!");
    }

    if (nonfinite) fprintf(stderr, "attention-hip: %ld non-finite result values\n", nonfinite);
    if (verbose) {
        struct sdpa_timing t;
        if (sdpa_last_timing(&t) == SDPA_OK)
            fprintf(stderr,
                    "attention-hip: m=%d n=%d dk=%d dv=%d gpus=%d q_batches=%d kv_splits=%d\n"
                    "attention-hip: total %.1f us | kv stage %.1f us | pipeline %.1f us | fused kernel %.1f us\n"
                    "attention-hip: max |result - answer| = %.3e\n",
                    m, n, dk, dv, t.n_gpus, t.q_batches, t.kv_splits, t.total_us, t.kv_stage_us,
                    t.pipeline_us, t.kernel_us, worst);
    }

    release_host_bufs();            /* before the engine goes away: pinned memory is the runtime's */
    sdpa_shutdown();
    return 0;
}
