/*
 * sdpa_oracle.c -- CPU restatement of the reference attention path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the
 * __graft_entry__.smoke() check and bench.py's cpu_baseline leg may load this
 * library.  The product path (libsdpa_hip.so + the package host code) never
 * links, imports or falls back to anything in oracle/.
 *
 * Parity status: PINNED.  oracle_attention_f64() is checked bit-for-bit
 * against the reference's own serial program (compiled unmodified from
 * /root/reference/attention.c into oracle/_ref/, see oracle/Makefile) on every
 * fixture in tests/golden/ (tests/test_oracle.py), and the fixtures themselves
 * were produced by that reference build (oracle/make_golden.py).
 * oracle_attention_sharded_f32() -- the fp32 K/V-sharded pipeline -- is checked
 * bit-for-bit against raw outputs of the reference's own MPI program
 * (attention-mpi.c unmodified, documented build flags, mpiexec -n 1/2/8,
 * oracle/ref_mpi_dump.c + oracle/make_ref_mpi_fp32.py ->
 * tests/golden/ref_mpi_fp32/).
 *
 * Every function cites the reference lines it restates; paths are relative to
 * /root/reference.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* K/V row partition: attention-mpi.c:19-27 (owner_count / owner_disp).       */
/* ------------------------------------------------------------------------- */
int oracle_owner_count(int n, int size, int rank)
{
    int q = n / size, r = n % size;
    return rank < r ? q + 1 : q;
}

int oracle_owner_disp(int n, int size, int rank)
{
    int q = n / size, r = n % size;
    return rank * q + (rank < r ? rank : r);
}

/* ------------------------------------------------------------------------- */
/* Serial fp64 attention: attention.c:20-75.                                  */
/* Three passes per query row: scores (dk-ascending dot, then *scale),        */
/* max-subtracted exp + sum, divide, then P.V accumulated j-ascending for     */
/* every output column.  Same operation order as the reference so the result  */
/* is bit-identical when compiled without value-changing optimisations.       */
/* ------------------------------------------------------------------------- */
void oracle_attention_f64(const double *Q, const double *K, const double *V,
                          double *result, int m, int n, int dk, int dv)
{
    const double scale = 1.0 / sqrt((double)dk);          /* attention.c:23 */
#pragma omp parallel
    {
        double *w = (double *)malloc(sizeof(double) * (size_t)n); /* :26 */
#pragma omp for schedule(static)
        for (int i = 0; i < m; ++i) {
            const double *q = Q + (size_t)i * dk;
            for (int j = 0; j < n; ++j) {                 /* :33-42 */
                const double *k = K + (size_t)j * dk;
                double acc = 0.0;
                for (int t = 0; t < dk; ++t) acc += q[t] * k[t];
                w[j] = acc * scale;
            }
            double top = w[0];                            /* :47-50 */
            for (int j = 1; j < n; ++j) if (w[j] > top) top = w[j];
            double denom = 0.0;                           /* :52-56 */
            for (int j = 0; j < n; ++j) { w[j] = exp(w[j] - top); denom += w[j]; }
            for (int j = 0; j < n; ++j) w[j] /= denom;    /* :57-59 */
            double *out = result + (size_t)i * dv;        /* :65-71 */
            for (int d = 0; d < dv; ++d) {
                double acc = 0.0;
                for (int j = 0; j < n; ++j) acc += w[j] * V[(size_t)j * dv + d];
                out[d] = acc;
            }
        }
        free(w);
    }
}

/* ------------------------------------------------------------------------- */
/* fp64 <-> fp32 array converts: attention-mpi.c:31-64 / :68-101.             */
/* _mm512_cvtpd_ps rounds to nearest-even under the default MXCSR, which is   */
/* what a C (float) cast does; the widening convert is exact.                 */
/* ------------------------------------------------------------------------- */
void oracle_cvt_d2f(float *dst, const double *src, size_t count)
{
    for (size_t i = 0; i < count; ++i) dst[i] = (float)src[i];
}

void oracle_cvt_f2d(double *dst, const float *src, size_t count)
{
    for (size_t i = 0; i < count; ++i) dst[i] = (double)src[i];
}

/* ------------------------------------------------------------------------- */
/* dot_avx512, attention-mpi.c:103-121, lane for lane in plain C: four         */
/* accumulators of 16 fp32 lanes fed by FMAs over 64-element strides, a        */
/* 16-wide clean-up loop and a masked tail into accumulator 0 (:109-119),      */
/* (acc0+acc1)+(acc2+acc3) per lane, then _mm512_reduce_add_ps -- gcc's        */
/* avx512fintrin.h reduces 16 -> 8 -> 4 -> 2 -> 1 by adding the upper half to  */
/* the lower half each time.  fmaf() is the one-rounding FMA of                */
/* _mm512_fmadd_ps, so the result is bit-identical to the reference's.         */
/* ------------------------------------------------------------------------- */
static float oracle_dot_lanes(const float *a, const float *b, int n)
{
    float acc[4][16];
    for (int g = 0; g < 4; ++g)
        for (int l = 0; l < 16; ++l) acc[g][l] = 0.0f;
    int i = 0;
    for (; i + 63 < n; i += 64)
        for (int g = 0; g < 4; ++g)
            for (int l = 0; l < 16; ++l)
                acc[g][l] = fmaf(a[i + 16 * g + l], b[i + 16 * g + l], acc[g][l]);
    for (; i + 15 < n; i += 16)
        for (int l = 0; l < 16; ++l) acc[0][l] = fmaf(a[i + l], b[i + l], acc[0][l]);
    for (int l = 0; l < n - i; ++l) acc[0][l] = fmaf(a[i + l], b[i + l], acc[0][l]);
    float v[16];
    for (int l = 0; l < 16; ++l) v[l] = (acc[0][l] + acc[1][l]) + (acc[2][l] + acc[3][l]);
    for (int l = 0; l < 8; ++l) v[l] = v[l + 8] + v[l];
    for (int l = 0; l < 4; ++l) v[l] = v[l + 4] + v[l];
    for (int l = 0; l < 2; ++l) v[l] = v[l] + v[l + 2];
    return v[0] + v[1];
}

/* ------------------------------------------------------------------------- */
/* One query row against one K/V shard, streaming online softmax:             */
/* attention-mpi.c:168-189.  Returns the UN-normalised contribution and the   */
/* shard-local (max, sum).  rmax starts at -inf (:172), contrib is zeroed     */
/* (:173), the running rescale is applied for every j>0 (:181, a plain        */
/* multiply, memset_zero_scale :142-166), the accumulate is one FMA per        */
/* element (axpy_avx512 :123-140).  Operation for operation the reference's:  */
/* tests/test_oracle.py checks it BIT FOR BIT against outputs of the          */
/* reference's own MPI program (tests/golden/ref_mpi_fp32/).                  */
/* ------------------------------------------------------------------------- */
void oracle_online_row_f32(float *contrib, float *lmax, float *lsum,
                           const float *q, const float *Kloc, const float *Vloc,
                           int n_local, int dk, int dv, float scale)
{
    float run_max = -INFINITY, run_sum = 0.0f;
    for (int d = 0; d < dv; ++d) contrib[d] = 0.0f;
    for (int j = 0; j < n_local; ++j) {
        float s = oracle_dot_lanes(q, Kloc + (size_t)j * dk, dk) * scale;   /* :176 */
        float prev = run_max;
        if (s > run_max) run_max = s;                      /* :178 */
        float fix = expf(prev - run_max);                  /* :179 */
        float p = expf(s - run_max);
        run_sum = run_sum * fix + p;                       /* :180 */
        const float *v = Vloc + (size_t)j * dv;
        if (j > 0) for (int d = 0; d < dv; ++d) contrib[d] *= fix;          /* :181 */
        for (int d = 0; d < dv; ++d) contrib[d] = fmaf(p, v[d], contrib[d]); /* :182 */
    }
    *lmax = run_max;
    *lsum = run_sum;
}

/* ------------------------------------------------------------------------- */
/* The sharded fp32 pipeline of attention-mpi.c:191-407 simulated in one      */
/* process for `parts` ranks:                                                 */
/*   - K,V converted to fp32 once (:224-225), Q per row (:303,:325)           */
/*   - shard r owns rows [owner_disp, +owner_count) (:199, :236-238)          */
/*   - per row: local triple (:333-338); gmax = max_r lmax (:342);            */
/*     corr = expf(lmax-gmax), lsum*=corr, contrib*=corr (:346-351);          */
/*     gsum = sum_r lsum (:354); inv = gsum==0 ? 0 : 1/gsum, contrib*=inv     */
/*     (:358-362); result = sum_r contrib (:380), widened to fp64 (:373,:396) */
/* The MPI reduction order is implementation defined; both sums here use the  */
/* pairwise tree of MPICH's recursive doubling / binomial reduce              */
/* (((0+1)+(2+3))+((4+5)+(6+7)) for 8 ranks), which is what the fixtures of   */
/* tests/golden/ref_mpi_fp32/ were produced with (MPICH 3.3.2).               */
/* scale = 1/sqrtf((float)dk) (:208).                                         */
/* ------------------------------------------------------------------------- */
static float tree_sum(const float *x, int stride, int lo, int hi)   /* ranks [lo, hi) */
{
    if (hi - lo == 1) return x[(size_t)lo * stride];
    int half = 1;
    while (half * 2 < hi - lo) half *= 2;
    return tree_sum(x, stride, lo, lo + half) + tree_sum(x, stride, lo + half, hi);
}

void oracle_attention_sharded_f32(const double *Q, const double *K, const double *V,
                                  double *result, int m, int n, int dk, int dv,
                                  int parts)
{
    float *Kf = (float *)malloc(sizeof(float) * (size_t)n * dk);
    float *Vf = (float *)malloc(sizeof(float) * (size_t)n * dv);
    oracle_cvt_d2f(Kf, K, (size_t)n * dk);
    oracle_cvt_d2f(Vf, V, (size_t)n * dv);
    const float scale = 1.0f / sqrtf((float)dk);
#pragma omp parallel
    {
        float *qf = (float *)malloc(sizeof(float) * (size_t)dk);
        float *part = (float *)malloc(sizeof(float) * (size_t)parts * dv);
        float *pmax = (float *)malloc(sizeof(float) * (size_t)parts);
        float *psum = (float *)malloc(sizeof(float) * (size_t)parts);
        float *tot = (float *)malloc(sizeof(float) * (size_t)dv);
#pragma omp for schedule(static)
        for (int i = 0; i < m; ++i) {
            oracle_cvt_d2f(qf, Q + (size_t)i * dk, (size_t)dk);
            for (int r = 0; r < parts; ++r) {
                int cnt = oracle_owner_count(n, parts, r);
                int off = oracle_owner_disp(n, parts, r);
                oracle_online_row_f32(part + (size_t)r * dv, pmax + r, psum + r, qf,
                                      Kf + (size_t)off * dk, Vf + (size_t)off * dv,
                                      cnt, dk, dv, scale);
            }
            float gmax = pmax[0];
            for (int r = 1; r < parts; ++r) if (pmax[r] > gmax) gmax = pmax[r];
            for (int r = 0; r < parts; ++r) {
                float fix = expf(pmax[r] - gmax);          /* :347 */
                psum[r] *= fix;
                for (int d = 0; d < dv; ++d) part[(size_t)r * dv + d] *= fix;
            }
            float gsum = tree_sum(psum, 1, 0, parts);      /* :354 */
            float inv = (gsum == 0.0f) ? 0.0f : 1.0f / gsum;
            for (int r = 0; r < parts; ++r)
                for (int d = 0; d < dv; ++d) part[(size_t)r * dv + d] *= inv;   /* :358-362 */
            for (int d = 0; d < dv; ++d) tot[d] = tree_sum(part + d, dv, 0, parts);   /* :380 */
            oracle_cvt_f2d(result + (size_t)i * dv, tot, (size_t)dv);
        }
        free(qf); free(part); free(pmax); free(psum); free(tot);
    }
    free(Kf); free(Vf);
}

/* Shard-local triples for ALL rows of one shard (fp32 in, fp32 out); used by  */
/* the world_size-2 gloo tests as the stand-in compute engine and by the GPU   */
/* tests to check the partial (contrib, lmax, lsum) outputs directly.          */
void oracle_shard_partial_f32(const float *Qf, const float *Kloc, const float *Vloc,
                              float *contrib, float *lmax, float *lsum,
                              int m, int n_local, int dk, int dv)
{
    const float scale = 1.0f / sqrtf((float)dk);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i)
        oracle_online_row_f32(contrib + (size_t)i * dv, lmax + i, lsum + i,
                              Qf + (size_t)i * dk, Kloc, Vloc, n_local, dk, dv, scale);
}

int oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
