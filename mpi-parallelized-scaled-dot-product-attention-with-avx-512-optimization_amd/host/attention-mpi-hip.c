/*
 * attention-mpi-hip.c -- the MPI-flavour drop-in: the reference's attention-mpi.c with the body
 * of its attention() replaced by the MI355X engine behind include/sdpa_hip.h.
 *
 * Contract kept (paths relative to the reference tree):
 *   boundary                      attention-mpi.c:191-192  void attention(double*,double*,double*,
 *                                 double*,int,int,int,int,int mpi_rank,int mpi_size)
 *   main()                        attention-mpi.c:497-541: MPI_Init; ONLY rank 0 reads the file and
 *                                 owns Q/K/V/result (the others pass NULL pointers and
 *                                 uninitialised dims, :508-517); MPI_Wtime around attention();
 *                                 MPI_Reduce(MAX) of the durations; rank 0 verifies and prints
 *   stdout                        "Correct!\nElapsed time: %.2lf us\n" or "Wrong!\n" (:526-531)
 *
 * Who computes: the reference shards K/V over MPI ranks because its workers are CPU cores.  Here
 * the workers are the node's GPUs, and ONE process drives all of them (P streams + RCCL over xGMI
 * inside sdpa_attention_f64).  So rank 0 calls the engine and ranks > 0 have nothing to add: their
 * attention() returns at once, they never create a HIP context, and the template's
 * MPI_Reduce(MAX) picks rank 0's duration.  `mpiexec -n 1` and `mpiexec -n 64` print the same
 * thing -- a submission script written for the reference keeps working.
 *
 * Built by mpicc (gcc underneath); sees nothing but the C header.  Environment as
 * attention-hip.c (SDPA_GPUS, SDPA_PLAN, SDPA_MERGE, SDPA_VERBOSE, $SDPA_DEBUG time_init, $SDPA_DEBUG pinned_io).
 */
#define _POSIX_C_SOURCE 200809L   /* clock_gettime under -std=c11 */
#include <mpi.h>

#include "sdpa_cli.h"

/* a fatal error on rank 0 ends the whole job: ranks > 0 are blocked in the template's MPI_Reduce */
static void abort_job(int code)
{
    MPI_Abort(MPI_COMM_WORLD, code);
}

/* ---- the drop-in boundary ------------------------------------------------ */
void attention(double *Q, double *K, double *V, double *result,
               int m, int n, int dk, int dv, int mpi_rank, int mpi_size)
{
    (void)mpi_size;
    if (mpi_rank != 0) return;          /* dims and pointers are only valid on rank 0 (:508-517) */
    die_if(sdpa_attention_f64(Q, K, V, result, m, n, dk, dv, SDPA_F_DEFAULT),
           "sdpa_attention_f64");
}

int main(int argc, char **argv)
{
    cli_name = "attention-mpi-hip";
    if (argc < 2) {
        fprintf(stderr, "Usage: %s <testing data>\n", argv[0]);
        return 1;
    }
    int rank, size;
    MPI_Init(&argc, &argv);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    cli_fail = abort_job;

    const double t_start = now_ms();
    const bool verbose = getenv("SDPA_VERBOSE") != NULL;
    const bool time_init = sdpa_debug_int("time_init", 0) != 0;

    struct problem p = {{0, 0, 0, 0}, NULL, NULL, NULL};
    double *result = NULL;
    int m = 0, n = 0, dk = 0, dv = 0;
    if (rank == 0) {
        /* engine before the read (page-locked reader memory), bad input reported first: as
         * attention-hip.c */
        if (!time_init) {
            precheck_file(argv[1]);
            die_if(cli_engine_up(), "sdpa_init");
            note_unused_gpus();
            if (size > 1)       /* the reference would compute on every rank; here they wait */
                fprintf(stderr, "%s: %d MPI ranks: rank 0 drives the GPU(s), ranks 1..%d only take part in the "
                        "template's MPI_Reduce\n", cli_name, size, size - 1);
            use_pinned = sdpa_debug_int("pinned_io", 1) != 0;
        }
        load_problem(argv[1], &p);
        m = p.dim[0]; n = p.dim[1]; dk = p.dim[2]; dv = p.dim[3];
        result = host_doubles((size_t)m * (size_t)dv);
        if (!result) {
            fprintf(stderr, "%s: out of memory\n", cli_name);
            cli_exit(1);
        }
        if (!time_init) die_if(sdpa_prepare(m, n, dk, dv, SDPA_F_DEFAULT), "sdpa_prepare");
    }

    double beg, duration, duration_max = 0.0;
    beg = MPI_Wtime();
    attention(p.q, p.k, p.v, result, m, n, dk, dv, rank, size);
    duration = MPI_Wtime() - beg;
    MPI_Reduce(&duration, &duration_max, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);

    if (rank == 0) {
        double worst = 0.0;
        long nonfinite = 0;
        if (check_answer(argv[1], result, &worst, &nonfinite))
            printf("Correct!\nElapsed time: %.2lf us\n", duration_max * 1e6);
        else
            puts("Wrong!");
        fflush(stdout);
        if (nonfinite) fprintf(stderr, "%s: %ld non-finite result values\n", cli_name, nonfinite);
        if (verbose) {
            fprintf(stderr, "%s: %d MPI ranks (rank 0 drives the GPUs); start -> result checked %.1f ms\n", cli_name,
                    size, now_ms() - t_start);
            report_verbose(m, n, dk, dv, worst);
        }
        release_host_bufs();        /* before the engine goes away: pinned memory is the runtime's */
        sdpa_shutdown();
    }
    MPI_Finalize();
    return 0;
}
